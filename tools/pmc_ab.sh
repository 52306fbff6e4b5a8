#!/bin/bash
# usage: tools/pmc_ab.sh <tag> "<counters>" [ab_variants.py args]   (run on the GPU box from the repo root)
# One rocprofv3 counter pass over tools/ab_variants.py; mean counter values per dispatch of the WCSPH pass kernels.
set -e
TAG=$1; CNT=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc $CNT --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/ab_variants.py --steps 2 "$@" > $OUT.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT density force > $OUT.txt
rm -rf $OUT
