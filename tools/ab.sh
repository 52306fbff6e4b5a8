#!/bin/bash
# usage: tools/ab.sh "<bench args>" v0 v1 ...   prints per-variant kernel times
ARGS=$1; shift
for v in "$@"; do
  timeout -k 10 300 python -u bench.py $ARGS --variant $v --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "variant $v failed"; tail -5 gpurun_out/ab_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json"))
k=d["kernels"]
print("variant $v: %.3g p-steps/s  %.3f ms/step | "%(d["value"], d["ms_per_step"]) + " ".join("%s=%.3f"%(n,k[n]["avg_ms"]) for n in k))
PY
done
