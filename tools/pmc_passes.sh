#!/bin/bash
# PMC passes over one 16-spp step of a workload (tools/prof_render.py): HBM traffic, L2 hit rate, VALU issue and lane use, where the waves' time goes.
# usage: bash tools/pmc_passes.sh <tag> <variant>     -> gpurun_out/<tag>/{fetch,write,l2,sq1,sq2,sqw,tcp}/ and summary.json
set -e -o pipefail
tag=$1; variant=$2
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; echo "[pass $name]"; timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o p -- python3 "$root/tools/prof_render.py" "$variant" 1 > "$out/$name.log" 2>&1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES
pass sq2 SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
pass sqw SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
pass tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -o p -- python3 "$root/tools/prof_render.py" "$variant" 1 > "$out/trace.log" 2>&1
cd "$root"
python3 tools/pmc_table.py "$out" > "$out/summary.json"
cat "$out/summary.json"
