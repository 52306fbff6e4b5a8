#!/bin/bash
# SQ counter pass of the headline bench (run on the GPU box from the repo root): VALU / LDS instruction counts and
# wave cycles per dispatch -> gpurun_out/prof/pmc_sq.txt  (profiles/valu.json is filled from its force and density rows)
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp FSA_BENCH_NO_DEVELOPED=1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline "$@" > $OUT/pmc_sq.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/pmc_sq > $OUT/pmc_sq.txt
rm -rf $OUT/pmc_sq
