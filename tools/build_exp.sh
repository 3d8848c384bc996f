#!/bin/bash
# Experiment builds of the library: bash tools/build_exp.sh <name> [extra compiler flags]  ->  exp/libenigma_<name>.so
# (PathIntegrator on triangle scenes only: ENIGMA_EXPERIMENT_BUILD; run with tools/exp_lib.py or ENIGMA_LIB=<path> bench.py)
set -e -o pipefail
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/exp" /tmp/expbuild
cd "$root/raytrace_amd/csrc"
make -s build/bvh_build.o build/sobol_host.o build/spectrum_host.o
hipcc --offload-arch=gfx950 -std=c++17 ${ENIGMA_EXP_OPT:--O1} -fPIC -ffp-contract=off -Wno-unused-variable -DENIGMA_EXPERIMENT_BUILD "$@" -c -o /tmp/expbuild/$name.o enigma_hip.hip
hipcc --offload-arch=gfx950 -shared -o "$root/exp/libenigma_$name.so" /tmp/expbuild/$name.o build/bvh_build.o build/sobol_host.o build/spectrum_host.o
echo "built exp/libenigma_$name.so"
