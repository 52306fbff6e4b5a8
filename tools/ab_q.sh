#!/bin/bash
# variant 3 with several quorum values of the replay loop (FSA_FAST_DBG = quorum << 4); args: extra ab_variants.py args
for q in 1 24 32 40 48 56; do
  echo "quorum=$q"; FSA_FAST_DBG=$((q*16)) timeout -k 10 150 python tools/ab_variants.py --variants 3 --steps 5 "$@" 2>&1 | grep "^3" || exit 1
done
