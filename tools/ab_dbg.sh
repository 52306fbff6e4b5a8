#!/bin/bash
# times variant 3 with experiment flags (FSA_FAST_DBG); args: flag values, then -- and extra ab_variants.py args
flags=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do flags+=("$1"); shift; done
[ "$1" == "--" ] && shift
[ ${#flags[@]} -eq 0 ] && flags=(0 8 1 2 9)
for d in "${flags[@]}"; do
  echo "dbg=$d"; FSA_FAST_DBG=$d timeout -k 10 100 python tools/ab_variants.py --variants 3 --steps 5 "$@" 2>&1 | grep "^3" || exit 1
done
