#!/bin/bash
# usage: tools/abl.sh "<bench args>" a0 a1 ...   (FSA_ABLATE values)
ARGS=$1; shift
for a in "$@"; do
  FSA_ABLATE=$a timeout -k 10 300 python -u bench.py $ARGS --no-cpu-baseline > gpurun_out/abl_$a.json 2> gpurun_out/abl_$a.err || { echo "ablate $a failed"; tail -5 gpurun_out/abl_$a.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abl_$a.json"))
k=d["kernels"]
print("ablate $a: " + " ".join("%s=%.3f"%(n,k[n]["avg_ms"]) for n in k if n in ("density","force")))
PY
done
