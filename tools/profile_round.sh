#!/bin/bash
# Round profile of the headline bench (run on the GPU box from the repo root):
#   1. bench.py as the driver runs it                     -> gpurun_out/prof/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same run   -> gpurun_out/prof/stats/
#   3. two --pmc passes (FETCH_SIZE, WRITE_SIZE), each a run of its own -> gpurun_out/prof/pmc_{fetch,write}/
# usage: tools/profile_round.sh [extra bench.py arguments]
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/bench.err
export FSA_BENCH_NO_DEVELOPED=1  # the profiled runs time the headline state only
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline "$@" > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline "$@" > $OUT/pmc_write.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/pmc_fetch > $OUT/pmc_fetch.txt
python3 $R/tools/pmc_summary.py $OUT/pmc_write > $OUT/pmc_write.txt
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# the raw traces are large; only the summaries travel back
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
