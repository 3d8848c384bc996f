#!/bin/bash
# Vector-memory counters of one 16-spp step (tools/prof_render.py): how busy the texture addresser / vector L1 are, where waves wait.
# usage: bash tools/pmc_mem.sh <tag> <variant>   -> gpurun_out/<tag>/{ta,tcp1,tcp2,sqw}/
set -e -o pipefail
tag=$1; variant=$2
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 60 rocprofv3 -L > "$out/counters.txt" 2>&1 || true
pass() { name=$1; shift; timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o p -- python3 "$root/tools/prof_render.py" "$variant" 1 > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -n 5 "$out/$name.log"; }; echo "pass $name done"; }
pass ta TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN2_sum
pass sqw SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
pass sqi SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES
