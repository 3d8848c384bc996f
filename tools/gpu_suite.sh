#!/bin/bash
# The GPU test suite and the default bench line, as the driver runs them at the end of a round: bash tools/gpu_suite.sh <tag> [bench arguments]
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
( while true; do sleep 50; echo "[heartbeat] $(date +%T)"; done ) &
hb=$!
trap "kill $hb 2>/dev/null" EXIT
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -v --timeout 600 --timeout-method thread > "$out/pytest_gpu.log" 2>&1 || { tail -40 "$out/pytest_gpu.log"; exit 1; }
tail -3 "$out/pytest_gpu.log"
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { tail -20 "$out/smoke.log"; exit 1; }
tail -1 "$out/smoke.log" | cut -c1-200
timeout -k 10 400 python3 bench.py "$@" > "$out/bench_default.json" 2> "$out/bench_default.err" || { tail -20 "$out/bench_default.err"; exit 1; }
python3 - "$out/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def brief(x, name):
    r = x["roofline"]
    print("%-8s %.2f ms/step %.0f Mrays/s %.0f Msamples/s job %.2f s (measured %s) bvh %.0f ms | closest %.3f ms bound %s frac %.2f | shadow %.3f | shade %.3f" % (
        name, x["ms_per_step"], x["value"], x["msamples_per_s"], x["job_seconds"], x["job_measured"], x["ms_bvh_build"], r["avg_launch_ms"], r["bound"], r["frac"],
        x.get("roofline_shadow", {}).get("avg_launch_ms", 0), x.get("roofline_shade", {}).get("avg_launch_ms", 0)), "cpu", x.get("cpu_baseline", {}).get("value"), "mre", x.get("mre_vs_cpu"))
brief(d, "cornell")
if "bigmesh" in d: brief(d["bigmesh"], "bigmesh")
PY
