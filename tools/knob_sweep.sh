# usage: bash tools/knob_sweep.sh <out dir> <ENV_NAME> <values...>: one short BigMesh bench per value of an experiment knob
out=$1; name=$2; shift 2
mkdir -p gpurun_out/$out
for v in "$@"; do
  env $name=$v timeout -k 10 120 python bench.py --variant bigmesh --no-cpu --steps 4 --warmup 1 > gpurun_out/$out/${name}_$v.json 2>/dev/null || exit 1
done
