#!/bin/bash
# usage: tools/pmc_iisph.sh <tag> "<counters>" [extra iisph_time.py args]   (run on the GPU box from the repo root)
set -e
TAG=$1; CNT=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc $CNT --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/iisph_time.py --dims 128 128 128 --steps 1 "$@" > $OUT.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT list_ tuned_pass > $OUT.txt
