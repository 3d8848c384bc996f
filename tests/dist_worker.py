"""Worker of the world_size-2 gloo test: the N>1 path of bench.py with the CPU oracle standing in
for the GPU renderer (same sharding helper, same film-reduce helper)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist

import common
import orc
from raytrace_amd import distributed


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spp, sampler = int(os.environ.get("DIST_SPP", "8")), os.environ.get("DIST_SAMPLER", "sobol")   # halton: any sample count, shared out unevenly
    sc, cam, film, integ = common.cornell_setup(res=32, spp=spp, variant="microfacet")
    o = orc.Oracle(sc, cam, film, integ, sampler=sampler)
    lo, hi = distributed.shard_range(0, spp, rank, world)
    o.render(lo, hi, threads=2)
    t = torch.from_numpy(o.film_xyzw().copy())
    distributed.reduce_film(t, dst=0)
    if rank == 0:
        full = orc.Oracle(sc, cam, film, integ, sampler=sampler)
        full.render(0, spp, threads=2)
        ref = full.film_xyzw()
        got = t.numpy()
        ok = np.allclose(got, ref, rtol=1e-5, atol=1e-6) and np.array_equal(got[..., 3], ref[..., 3])
        open(sys.argv[1], "w").write("ok" if ok else "mismatch %g" % np.abs(got - ref).max())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
