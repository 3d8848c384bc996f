#!/bin/bash
# VolPathIntegrator tests + the two fog benches: bash tools/vol_check.sh <tag>
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$1
mkdir -p "$out"
cd "$root"
( while true; do sleep 50; echo "[heartbeat] $(date +%T)"; done ) &
hb=$!
trap "kill $hb 2>/dev/null" EXIT
timeout -k 10 600 python -u -m pytest tests/test_gpu_volpath.py tests/test_gpu_random_scenes.py -m gpu -x -v --timeout 600 --timeout-method thread > "$out/pytest_vol.log" 2>&1 || { tail -40 "$out/pytest_vol.log"; exit 1; }
tail -2 "$out/pytest_vol.log"
for v in fog bigmesh_fog; do
  timeout -k 10 280 python3 bench.py --variant $v --integrator volpath --steps 6 --warmup 2 --no-cpu > "$out/bench_$v.json" 2> "$out/bench_$v.err" || { tail -5 "$out/bench_$v.err"; exit 1; }
  python3 - "$out/bench_$v.json" $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-12s %.2f ms/step %.0f Mrays/s %.0f Msamples/s | closest %.3f x%d | shadow %.3f | shade %.3f" % (sys.argv[2], d["ms_per_step"], d["value"], d["msamples_per_s"],
      d["roofline"]["avg_launch_ms"], d["roofline"]["launches"], d["roofline_shadow"]["avg_launch_ms"], d["roofline_shade"]["avg_launch_ms"]))
PY
done
