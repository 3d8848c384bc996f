#!/bin/bash
# Per-kernel times of a few 16-spp steps of a workload (tools/prof_render.py) under rocprofv3 --kernel-trace.
# usage: bash tools/kstats_run.sh <tag> <variant> [steps]   -> gpurun_out/<tag>/kstats_<variant>.txt
set -e -o pipefail
tag=$1; variant=$2; steps=${3:-3}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace_$variant" -o p -- python3 "$root/tools/prof_render.py" "$variant" "$steps" > "$out/trace_$variant.log" 2>&1
cd "$root"
python3 tools/kstats.py "$(ls $out/trace_$variant/*kernel_trace.csv | head -n 1)" "$steps" > "$out/kstats_$variant.txt"
head -n 16 "$out/kstats_$variant.txt"
