#!/bin/bash
# The measurement set kept under profiles/: bench lines of every workload, kernel stats and PMC passes.
# usage: bash tools/final_set.sh <tag>      -> gpurun_out/<tag>/
set -e -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout -k 10 280 python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
echo default done
b() { name=$1; shift; timeout -k 10 200 python3 bench.py --no-cpu "$@" > "$out/bench_$name.json" 2> "$out/bench_$name.err"; echo "$name done"; }
b bigmesh_path --variant bigmesh
b manylights_path --variant manylights
b manylights_path_spatial --variant manylights --light-strategy spatial
b microfacet_direct-all --integrator direct-all
b manylights_direct-one --variant manylights --integrator direct-one
b fog_volpath --variant fog --integrator volpath
b bigmesh_fog_volpath --variant bigmesh_fog --integrator volpath --steps 8
bash tools/profile_bench.sh "$tag/prof" > "$out/prof.log" 2>&1
echo prof done
bash tools/pmc_passes.sh "$tag/pmc_cornell" microfacet > "$out/pmc_cornell.log" 2>&1
echo pmc cornell done
bash tools/pmc_passes.sh "$tag/pmc_bigmesh" bigmesh > "$out/pmc_bigmesh.log" 2>&1
echo pmc bigmesh done
bash tools/pmc_passes.sh "$tag/pmc_bigmesh_fog" bigmesh_fog > "$out/pmc_bigmesh_fog.log" 2>&1
echo pmc bigmesh_fog done
for v in microfacet bigmesh bigmesh_fog; do bash tools/kstats_run.sh "$tag" $v 3 > /dev/null; echo "kstats $v done"; done
