#!/bin/bash
# usage (on the GPU box, from the repo root): tools/ab_steps.sh name1 name2 ...
# times the IISPH and DFSPH steps (AB_STEPS="dfsph" for one of them, AB_ARGS for more arguments) at 16 M particles with fluidsolver_amd/lib/variants/<name>.so ("base" = the normal build)
for v in "$@"; do
  if [ "$v" = base ]; then unset FSA_LIB_PATH; else export FSA_LIB_PATH=$PWD/fluidsolver_amd/lib/variants/$v.so; fi
  for t in ${AB_STEPS:-dfsph iisph}; do
    timeout -k 10 300 python -u tools/${t}_time.py --dims 256 256 256 --steps 10 ${AB_ARGS} > gpurun_out/ab_${t}_$v.json 2> gpurun_out/ab_${t}_$v.err || { echo "$v $t failed"; tail -5 gpurun_out/ab_${t}_$v.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab_${t}_$v.json").read().strip().splitlines()[-1])
k=d["kernels_ms_per_step"]
top=sorted(k.items(), key=lambda kv:-kv[1])[:5]
print("$v $t: %.2f ms/step | "%d["ms_per_step"] + " ".join("%s=%.2f"%(a,b) for a,b in top))
PY
  done
done
