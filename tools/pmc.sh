#!/bin/bash
# usage: tools/pmc.sh <tag> <variant> <scene> "<counters>"   (run on the GPU box from the repo root)
set -e
TAG=$1; VAR=$2; SCENE=$3; CNT=$4
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $CNT --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --scene $SCENE --variant $VAR --steps 2 --warmup 0 --spinup 4 --jitter 0.2 --no-cpu-baseline > $OUT.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT density force > $OUT.txt
