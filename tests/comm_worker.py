"""One rank of the two-GPU test of the library's own film reduce (tests/test_gpu_distributed.py): no torch, no launcher;
the 128-byte communicator id travels through a file."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

import common
import raytrace_amd as ra
from raytrace_amd import distributed


def main():
    rank, world, idfile, result = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    spp = 16
    sc, cam, film, integ = common.cornell_setup(res=64, spp=spp, variant="microfacet")
    r = ra.Renderer(rank).configure(sc, cam, film, integ)
    if rank == 0:
        ident = ra.Renderer.comm_unique_id()
        with open(idfile + ".part", "wb") as f:
            f.write(ident)
        os.rename(idfile + ".part", idfile)
    else:
        for _ in range(6000):
            if os.path.exists(idfile):
                break
            time.sleep(0.01)
        ident = open(idfile, "rb").read()
    r.comm_init(rank, world, ident)
    assert r.comm_info() == (rank, world)
    lo, hi = distributed.shard_range(0, spp, rank, world)
    r.render(lo, hi)
    r.film_reduce(0)
    if rank == 0:
        got = r.film()
        whole = ra.Renderer(0).configure(sc, cam, film, integ)
        whole.render(0, spp)
        ref = whole.film()
        ok = np.array_equal(got[..., 3], ref[..., 3]) and np.allclose(got, ref, rtol=1e-5, atol=1e-5)
        open(result, "w").write("ok" if ok else "mismatch %g" % np.abs(got - ref).max())
        whole.close()
    r.close()


if __name__ == "__main__":
    main()
