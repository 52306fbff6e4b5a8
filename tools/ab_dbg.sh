#!/bin/bash
# times variant 3 with ablation flags (FSA_FAST_DBG)
for d in 0 1 2 3 4 6; do
  echo "dbg=$d"; FSA_FAST_DBG=$d timeout -k 10 100 python tools/ab_variants.py --variants 3 --steps 5 2>&1 | grep "^3"
done
