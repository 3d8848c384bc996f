#!/bin/bash
# Round-3 first GPU call: the new tests, then per-kernel times of the strict build against a build with fast division,
# square roots and contraction everywhere (an upper bound of what relaxed arithmetic in the shade kernels can give).
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3a
mkdir -p "$out"
cd "$root"
# long CPU phases (the oracle building its tree over 1.18 M triangles) print nothing: a heartbeat keeps the run from looking hung;
# every GPU step below has a time limit of its own
( while true; do sleep 50; echo "[heartbeat] $(date +%T)"; done ) &
hb=$!
trap "kill $hb 2>/dev/null" EXIT
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_volpath.py -m gpu -x -v --timeout 600 --timeout-method thread -k "production_kernel or full_size or textured_big_mesh" > "$out/pytest_new.log" 2>&1 || { tail -30 "$out/pytest_new.log"; exit 1; }
tail -3 "$out/pytest_new.log"
cd /tmp && export TMPDIR=/tmp
for lib in strict fastall; do
  for v in microfacet bigmesh; do
    if [ $lib = strict ]; then L=$root/raytrace_amd/libenigma_hip.so; else L=$root/exp/libenigma_$lib.so; fi
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$out/trace_${lib}_$v" -o p -- python3 "$root/tools/exp_lib.py" "$L" $v 3 > "$out/trace_${lib}_$v.log" 2>&1
    python3 "$root/tools/kstats.py" $(find "$out/trace_${lib}_$v" -name "*kernel_trace.csv" | head -1) 3 > "$out/kstats_${lib}_$v.txt"
    echo "== $lib $v"; head -12 "$out/kstats_${lib}_$v.txt"; tail -1 "$out/trace_${lib}_$v.log"
  done
done
cd "$root"
timeout -k 10 280 python3 bench.py --variant bigmesh_fog --integrator volpath --steps 4 --warmup 1 --no-cpu > "$out/bench_bigmesh_fog.json" 2> "$out/bench_bigmesh_fog.err"
cat "$out/bench_bigmesh_fog.json"
