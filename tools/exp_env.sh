#!/bin/bash
# usage: bash tools/exp_env.sh <tag> <lib name> VAR=value [VAR=value ...]: short BigMesh and Cornell benches of one library, once plain and once per setting
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; name=$2; shift 2
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
if [ "$name" = product ]; then export ENIGMA_LIB=$root/raytrace_amd/libenigma_hip.so; else export ENIGMA_LIB=$root/exp/libenigma_$name.so; fi
for cfg in PLAIN=1 "$@"; do
  for v in bigmesh microfacet; do
    f=$out/bench_${name}_$(echo $cfg | tr '=' '_')_$v.json
    env $cfg timeout -k 10 200 python3 bench.py --variant $v --steps 6 --warmup 2 --no-cpu > "$f" 2> "${f%.json}.err" || { tail -5 "${f%.json}.err"; exit 1; }
    python3 - "$f" "$cfg" $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, s = d["roofline"], d.get("roofline_shadow", {})
print("%-24s %-10s %7.2f ms/step %8.0f Mrays/s | closest %.3f ms (%.2f nodes %.2f tris /ray, lanes %.2f, tris/step %.1f, wave-steps/ray %.1f) | shadow %.3f ms | shade %.3f ms" % (
    sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], r["avg_launch_ms"], r["nodes_per_ray"], r["triangles_per_ray"], r.get("lanes_busy_node_step") or 0,
    r.get("triangles_per_tri_step") or 0, r["wave_steps_per_ray"], s.get("avg_launch_ms", 0), d.get("roofline_shade", {}).get("avg_launch_ms", 0)))
PY
  done
done
