#!/bin/bash
# Measurement sweeps over the BigMesh bench: bash tools/bvh_sweep.sh <tag> VAR=value [VAR=value ...] runs one configuration
# per argument (ENIGMA_BVH_CPRIM, ENIGMA_BVH_BINS, ENIGMA_REFILL_IDLE ...) and prints step time, launch times and visits per ray.
out=gpurun_out/$1; shift
mkdir -p $out
run() { tag=$(echo "$1" | tr '=. ' '___'); env "$@" timeout -k 10 200 python bench.py --variant bigmesh --steps 4 --warmup 1 --no-cpu > $out/$tag.json 2>$out/$tag.err || return 1; python3 -c "
import json,sys
d=json.loads(open('$out/$tag.json').read().strip().splitlines()[-1])
r,s=d['roofline'],d['roofline_shadow']
print('$tag',round(d['ms_per_step'],2),'closest',round(r['avg_launch_ms'],3),round(r['nodes_per_ray'],2),round(r['triangles_per_ray'],2),r['lanes_busy_node_step'] and round(r['lanes_busy_node_step'],3),r['triangles_per_tri_step'] and round(r['triangles_per_tri_step'],1),'shadow',round(s['avg_launch_ms'],3),round(s['nodes_per_ray'],2),round(s['triangles_per_ray'],2))"; }
for cfg in "$@"; do run $cfg || exit 1; done
