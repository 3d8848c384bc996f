#!/bin/bash
# Kernel stats and HBM traffic counters of the default bench command, into gpurun_out/<tag>/.
# usage: bash tools/profile_bench.sh <tag> [bench.py arguments]
set -e -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- python3 "$root/bench.py" --steps 5 --warmup 1 --no-cpu "$@" > "$out/stats.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o bench -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu "$@" > "$out/fetch.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o bench -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu "$@" > "$out/write.log" 2>&1
cd "$root"
python3 tools/pmc_summary.py "$out/pmc.json" "$out/pmc_fetch" "$out/pmc_write" > /dev/null
ls "$out" "$out/stats"
