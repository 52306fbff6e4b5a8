#!/bin/bash
# usage (on the GPU box, from the repo root): tools/ab_libs.sh "<bench args>" name1 name2 ...
# times the headline bench with fluidsolver_amd/lib/variants/<name>.so ("base" = the normal build): kernel times
ARGS=$1; shift
export FSA_BENCH_NO_DEVELOPED=1
for v in "$@"; do
  if [ "$v" = base ]; then unset FSA_LIB_PATH; else export FSA_LIB_PATH=$PWD/fluidsolver_amd/lib/variants/$v.so; fi
  timeout -k 10 300 python -u bench.py $ARGS --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$v failed"; tail -5 gpurun_out/ab_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json"))
k=d["kernels"]
print("$v: %.3f ms/step | "%d["ms_per_step"] + " ".join("%s=%.3f"%(n,k[n]["avg_ms"]) for n in k if n in ("density","force","wall_force")))
PY
done
