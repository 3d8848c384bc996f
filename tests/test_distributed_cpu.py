"""world_size-2 gloo test of the sample-sharded render + film reduce (the multi-GPU path on CPU)."""
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


import pytest


@pytest.mark.parametrize("sampler,spp", [("sobol", 8), ("halton", 7)])
def test_two_rank_sharded_render_matches_single_process(tmp_path, sampler, spp):
    port = _free_port()
    result = tmp_path / "result.txt"
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", DIST_SAMPLER=sampler, DIST_SPP=str(spp))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(result)], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0
    assert result.read_text() == "ok"


def _group_is_empty(pgid):
    """No process of the session the launcher was started in is left."""
    try:
        os.killpg(pgid, 0)
    except ProcessLookupError:
        return True
    return False


def test_enigma_render_two_ranks_end_together_when_one_fails(tmp_path):
    """enigma_render --gpus 2: the parent forks its ranks before anything touches the GPU and hands the communicator id
    over through a file.  In this container no rank can get as far as a GPU call; what is checked is the launcher: the
    failure is reported, the exit code is non-zero, it takes seconds and not the minute a waiting rank would poll for,
    and no child is left behind.  (On a box where both ranks find a GPU the render succeeds and the file exists.)"""
    import time
    exe = os.path.join(ROOT, "raytrace_amd", "enigma_render")
    assert os.path.exists(exe), "raytrace_amd/enigma_render is not built (make -C raytrace_amd/csrc)"
    scene = tmp_path / "tiny.pbrt"
    scene.write_text("""LookAt 0 0 -5  0 0 0  0 1 0
Camera "perspective" "float fov" [40]
Film "image" "integer xresolution" [8] "integer yresolution" [8]
Sampler "sobol" "integer pixelsamples" [4]
Integrator "path" "integer maxdepth" [2] "string lightsamplestrategy" ["power"]
WorldBegin
LightSource "point" "rgb I" [10 10 10] "point from" [0 3 -2]
Material "matte"
Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-1 -1 0  1 -1 0  0 1 0]
WorldEnd
""")
    out = tmp_path / "out.pfm"
    t0 = time.time()
    p = subprocess.Popen([exe, "--gpus", "2", str(scene), str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
    try:
        _, err = p.communicate(timeout=120)
    finally:
        if p.poll() is None:
            os.killpg(p.pid, 9)
    took = time.time() - t0
    assert _group_is_empty(p.pid), "a rank outlived the launcher"
    assert not (tmp_path / "out.pfm.commid").exists()
    if p.returncode == 0:
        assert out.exists()
    else:
        assert took < 45, "the launcher waited for a rank that could never arrive"
        assert b"failed" in err and b"rank" in err


def test_bench_spawned_ranks_end_together_when_one_fails():
    """bench.py --gpus 2 started without a launcher spawns its own ranks and must end all of them, with a non-zero exit
    code, as soon as one fails (here: no GPU to put a rank on)."""
    env = dict(os.environ, ENIGMA_BENCH_DEADLINE_S="200")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"], cwd=ROOT, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
    try:
        out, err = p.communicate(timeout=280)
    finally:
        if p.poll() is None:
            os.killpg(p.pid, 9)
    assert _group_is_empty(p.pid), "a rank outlived bench.py"
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert p.returncode == 0, err[-2000:]
    else:
        assert p.returncode != 0 and b"ranks exited with" in err


def test_bench_rank_under_a_foreign_launcher_ends_itself_at_the_deadline():
    """Under torch.distributed.run nobody watches bench.py's ranks: each arms a timer that ends the process, with a
    message and exit code 3, when ENIGMA_BENCH_DEADLINE_S has passed (a peer lost inside a collective never returns)."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench; bench.arm_watchdog(); time.sleep(30)" % ROOT)
    env = dict(os.environ, ENIGMA_BENCH_DEADLINE_S="0.5", RANK="1")
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=60)
    assert p.returncode == 3 and b"rank 1 still running" in p.stderr
    assert time.time() - t0 < 25
