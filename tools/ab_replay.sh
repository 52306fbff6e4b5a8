#!/bin/bash
# usage (on the GPU box, from the repo root): tools/ab_replay.sh
# the DFSPH step at 16 M particles with and without the replay of the hit record
for v in 0 1; do
  FSA_HIT_REPLAY=$v timeout -k 10 300 python -u tools/dfsph_time.py --dims 256 256 256 --steps 10 > gpurun_out/ab_replay_$v.json 2> gpurun_out/ab_replay_$v.err || { echo "replay=$v failed"; tail -5 gpurun_out/ab_replay_$v.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_replay_$v.json").read().strip().splitlines()[-1])
k=d["kernels_ms_per_step"]
top=sorted(k.items(), key=lambda kv:-kv[1])[:8]
print("replay=$v: %.2f ms/step | "%d["ms_per_step"] + " ".join("%s=%.2f"%(a,b) for a,b in top))
PY
done
