#!/bin/bash
# usage: tools/profile_tool.sh <tag> <tools/script.py> [script args]   (run on the GPU box from the repo root)
# rocprofv3 --kernel-trace --stats of a timing tool; the kernel summary lands in gpurun_out/prof/<tag>_kernel_stats.csv
set -e
TAG=$1; TOOL=$2; shift 2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$TAG -- python3 $R/$TOOL "$@" > $OUT/stats_$TAG.log 2>&1
find $OUT/stats_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
rm -rf $OUT/stats_$TAG
