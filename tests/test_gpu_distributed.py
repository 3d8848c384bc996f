"""The pieces of the multi-GPU path that one GPU can exercise: the torch view of the library's device film is the
film itself (no copy), RCCL accepts it (memory the library allocated, not torch), and the sample-sharded render +
sum of films equals the whole render.  The N > 1 reduce itself runs in tests/test_distributed_cpu.py (gloo)."""
import os
import socket

import numpy as np
import pytest
import torch  # before the library: the PyTorch wheel brings its own HIP runtime, and the one that is loaded first serves the process

import common
import raytrace_amd as ra
from raytrace_amd import distributed

pytestmark = pytest.mark.gpu


def test_film_view_aliases_device_film_and_rccl_accepts_it():
    import torch
    import torch.distributed as dist
    sc, cam, film, integ = common.cornell_setup(res=64, spp=16, variant="microfacet")
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    dev = torch.device("cuda", 0)
    t = distributed.film_tensor(r, dev)
    assert t.numel() == 64 * 64 * 4 and float(t.abs().sum()) == 0.0
    r.render(0, 16)
    torch.cuda.synchronize()
    whole = r.film()
    assert np.array_equal(t.cpu().numpy().reshape(64, 64, 4), whole)         # the view sees what the library rendered
    # two "ranks" on this one GPU: disjoint sample ranges into two films, summed with torch on the device views
    lo0, hi0 = distributed.shard_range(0, 16, 0, 2)
    lo1, hi1 = distributed.shard_range(0, 16, 1, 2)
    assert (lo0, hi0, lo1, hi1) == (0, 8, 8, 16)
    r2 = ra.Renderer(0).configure(sc, cam, film, integ)
    r.clear(); r.render(lo0, hi0); r2.render(lo1, hi1)
    t2 = distributed.film_tensor(r2, dev)
    t += t2                                                                   # what reduce(SUM) does on rank 0
    torch.cuda.synchronize()
    summed = r.film()
    assert np.array_equal(summed[..., 3], whole[..., 3])
    assert np.allclose(summed, whole, rtol=1e-5, atol=1e-5)
    # RCCL on library-owned memory (world size 1: the collective is a no-op copy, but registration and launch happen)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        before = t.clone()
        dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        assert torch.equal(t, before)
    finally:
        dist.destroy_process_group()
    r.close(); r2.close()


def test_native_communicator_of_one_rank():
    """enigma_comm_* with a single rank: RCCL loads, the communicator reports itself, the reduce leaves the film as it is."""
    sc, cam, film, integ = common.cornell_setup(res=32, spp=4, variant="diffuse")
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    assert r.comm_info() == (0, 1)
    r.render(0, 4)
    before = r.film()
    r.film_reduce(0)                                   # no communicator: a no-op
    ident = ra.Renderer.comm_unique_id()
    assert len(ident) == 128 and any(ident)
    r.comm_init(0, 1, ident)
    assert r.comm_info() == (0, 1)
    r.film_reduce(0)
    assert np.array_equal(r.film(), before)
    with pytest.raises(ra.EnigmaError) as e:
        r.comm_init(0, 1, ident)                       # a handle joins one communicator
    assert e.value.code == -2
    with pytest.raises(ra.EnigmaError) as e:
        r.film_reduce(3)
    assert e.value.code == -1
    r.close()


def _visible_gpus():
    import torch
    return torch.cuda.device_count()


def test_two_rank_native_reduce_matches_single_render(tmp_path):
    """Two processes, two GPUs: each renders half of the sample numbers, enigma_film_reduce (RCCL inside the library)
    sums the films onto rank 0, which must hold what one process renders alone.  Skipped on a one-GPU box."""
    if _visible_gpus() < 2:
        pytest.skip("needs two GPUs")
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    idfile, result = tmp_path / "commid", tmp_path / "result.txt"
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "comm_worker.py"), str(rank), "2", str(idfile), str(result)]) for rank in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    assert result.read_text() == "ok"


def test_bench_two_ranks_rehearsed_on_one_gpu():
    """The N-rank control flow of bench.py on a box with one GPU: it spawns its two ranks, both on device 0, with gloo
    under torch.distributed.  RCCL refuses two ranks of one device, so the library's communicator fails on every rank,
    the ranks agree on the fallback and torch reduces the film.  Checks the line's accounting, not a speed."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ENIGMA_BENCH_ONE_GPU="1", ENIGMA_BENCH_BACKEND="gloo", ENIGMA_BENCH_ALLOW_FALLBACK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu", "--variant", "microfacet"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["config"]["spp_rendered"] == 2 * 2 * 16  # ranks x steps x 16 spp
    assert "torch.distributed reduce" in line["config"]["parallelism"] and line["rccl_ranks"] == 1
    assert line["reduce"] == "torch-fallback"            # said at the top level, so that nobody credits the library with torch's reduce
    per_sample = line["rays"]["per_sample"]
    assert 3.5 < per_sample < 4.0  # the Cornell box's rays per camera sample: both ranks' work is in the totals
    samples = 1920 * 1080 * line["config"]["spp_rendered"]
    assert abs(line["msamples_per_s"] * 1e6 * line["ms_per_step"] * 1e-3 * line["steps"] / samples - 1.0) < 1e-6


def test_bench_refuses_to_fall_back_unasked():
    """Without ENIGMA_BENCH_ALLOW_FALLBACK a run whose native communicator does not come up must not produce a number:
    the same two ranks on one GPU end with a non-zero exit code and a line that says why."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ENIGMA_BENCH_ONE_GPU="1", ENIGMA_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ENIGMA_BENCH_ALLOW_FALLBACK"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu", "--variant", "microfacet"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert run.returncode != 0
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["reduce"] == "torch-fallback" and line["value"] is None and "ENIGMA_BENCH_ALLOW_FALLBACK" in line["error"]
