#!/bin/bash
# times variant 3 with experiment flags (FSA_FAST_DBG); args: extra ab_variants.py args
for d in 0 8 1 2 9; do
  echo "dbg=$d"; FSA_FAST_DBG=$d timeout -k 10 100 python tools/ab_variants.py --variants 3 --steps 5 "$@" 2>&1 | grep "^3" || exit 1
done
