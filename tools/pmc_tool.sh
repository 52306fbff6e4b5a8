#!/bin/bash
# usage: tools/pmc_tool.sh <tag> "<counters>" <tools/script.py> [script args]   (run on the GPU box from the repo root)
# One rocprofv3 counter pass over a timing tool; mean counter values per dispatch of the pass kernels.
set -e
TAG=$1; CNT=$2; TOOL=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc $CNT --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/$TOOL --dims 128 128 128 --steps 1 "$@" > $OUT.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT list_ tuned_pass > $OUT.txt
