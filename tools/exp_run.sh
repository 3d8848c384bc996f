#!/bin/bash
# usage: bash tools/exp_run.sh <tag> <lib name> [<lib name> ...]   (exp/libenigma_<name>.so; "product" = the library in the tree)
# Per library: films / hits / ray counts of a short render (compared with the first library's) and short benches of both scenes.
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
first=""
for name in "$@"; do
  if [ "$name" = product ]; then lib=$root/raytrace_amd/libenigma_hip.so; else lib=$root/exp/libenigma_$name.so; fi
  export ENIGMA_LIB=$lib
  timeout -k 10 300 python3 tools/exp_film.py "$out/film_$name.npz" > "$out/film_$name.log" 2>&1 || { tail -5 "$out/film_$name.log"; exit 1; }
  if [ -z "$first" ]; then first=$name; else python3 tools/exp_film.py --compare "$out/film_$first.npz" "$out/film_$name.npz" || echo "== $name DIFFERS from $first"; fi
  for v in bigmesh microfacet; do
    timeout -k 10 200 python3 bench.py --variant $v --steps 6 --warmup 2 --no-cpu > "$out/bench_${name}_$v.json" 2> "$out/bench_${name}_$v.err" || { tail -5 "$out/bench_${name}_$v.err"; exit 1; }
    python3 - "$out/bench_${name}_$v.json" "$name" $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, s = d["roofline"], d.get("roofline_shadow", {})
print("%-10s %-10s %7.2f ms/step %8.0f Mrays/s | closest %.3f ms (%.2f nodes %.2f tris /ray, lanes %.2f, tris/step %.1f, wave-steps/ray %.1f) | shadow %.3f ms | shade %.3f ms" % (
    sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], r["avg_launch_ms"], r["nodes_per_ray"], r["triangles_per_ray"], r.get("lanes_busy_node_step") or 0,
    r.get("triangles_per_tri_step") or 0, r["wave_steps_per_ray"], s.get("avg_launch_ms", 0), d.get("roofline_shade", {}).get("avg_launch_ms", 0)))
PY
  done
done
