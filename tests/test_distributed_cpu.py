"""world_size-2 gloo test of the sample-sharded render + film reduce (the multi-GPU path on CPU)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharded_render_matches_single_process(tmp_path):
    port = _free_port()
    result = tmp_path / "result.txt"
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(result)], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0
    assert result.read_text() == "ok"
