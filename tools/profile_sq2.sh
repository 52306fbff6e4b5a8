#!/bin/bash
# Second SQ counter pass of the headline bench (run on the GPU box from the repo root): where the waves of the two
# neighbour passes spend their cycles (parked at waits, issue stalls, LDS array cycles and bank conflicts)
# -> gpurun_out/prof/pmc_sq2.txt
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp FSA_BENCH_NO_DEVELOPED=1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline "$@" > $OUT/pmc_sq2.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/pmc_sq2 > $OUT/pmc_sq2.txt
rm -rf $OUT/pmc_sq2
