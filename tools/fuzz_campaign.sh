#!/bin/bash
# A fuzzing campaign on the GPU box: bash tools/fuzz_campaign.sh <tag> <first seed> <count>
# (render seeds with the default and the HBM builds, scene files excluded; taps, BSDF points and adversarial rays at a tenth of the count)
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; first=$2; count=$3
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
( while true; do sleep 50; echo "[heartbeat] $(date +%T) $(tail -n 1 $out/render.log 2>/dev/null | cut -c1-100)"; done ) &
hb=$!
trap "kill $hb 2>/dev/null" EXIT
rc=0
timeout -k 10 900 python3 -u tools/fuzz_scenes.py $first $count "$out/render.json" > "$out/render.log" 2>&1 || rc=1
tail -n 3 "$out/render.log"
ENIGMA_NO_LDS_SCENE=1 timeout -k 10 600 python3 -u tools/fuzz_scenes.py $((first + count)) $((count / 3)) "$out/render_hbm.json" > "$out/render_hbm.log" 2>&1 || rc=1
tail -n 3 "$out/render_hbm.log"
FUZZ_REUSE=1 timeout -k 10 600 python3 -u tools/fuzz_scenes.py $((first + 2 * count)) $((count / 3)) "$out/render_reuse.json" > "$out/render_reuse.log" 2>&1 || rc=1
tail -n 3 "$out/render_reuse.log"
for mode in taps bsdf rays; do
  timeout -k 10 600 python3 -u tools/fuzz_scenes.py --$mode $first $((count / 10)) > "$out/$mode.log" 2>&1 || rc=1
  tail -n 2 "$out/$mode.log"
done
exit $rc
