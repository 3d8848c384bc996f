#!/bin/bash
# Copies the summaries of a measurement set (tools/final_set.sh <tag>) into profiles/ and rebuilds profiles/traffic.json.
# usage: bash tools/keep_set.sh <tag> <commit the set was measured on>
set -e -o pipefail
tag=$1; commit=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/gpurun_out/$tag; dst=$root/profiles
for f in "$src"/bench_*.json; do cp "$f" "$dst/${tag}_$(basename "$f")"; done
for v in microfacet bigmesh bigmesh_fog; do [ -f "$src/kstats_$v.txt" ] && cp "$src/kstats_$v.txt" "$dst/${tag}_kstats_$v.txt"; done
cp "$src/pmc_cornell/summary.json" "$dst/${tag}_pmc_cornell_per_kernel.json"
cp "$src/pmc_bigmesh/summary.json" "$dst/${tag}_pmc_bigmesh_per_kernel.json"
[ -f "$src/pmc_bigmesh_fog/summary.json" ] && cp "$src/pmc_bigmesh_fog/summary.json" "$dst/${tag}_pmc_bigmesh_fog_per_kernel.json"
stats=$(ls "$src"/prof/stats/*kernel_stats.csv 2>/dev/null | head -n 1)
[ -n "$stats" ] && cp "$stats" "$dst/${tag}_bench_kernel_stats.csv"
cd "$root"
python3 tools/traffic_json.py profiles/traffic.json --commit "$commit" \
  microfacet_path_power=profiles/${tag}_pmc_cornell_per_kernel.json bigmesh_path_power=profiles/${tag}_pmc_bigmesh_per_kernel.json \
  $( [ -f "$dst/${tag}_pmc_bigmesh_fog_per_kernel.json" ] && echo bigmesh_fog_volpath_power=profiles/${tag}_pmc_bigmesh_fog_per_kernel.json ) > /dev/null
ls "$dst" | grep "^$tag" 
