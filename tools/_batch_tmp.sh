export FSA_BENCH_NO_DEVELOPED=1
for r in 6 4 6 4; do
  echo "run_log $r" >> gpurun_out/s5_rl2.txt
  FSA_XCD_RUN_LOG=$r timeout -k 10 300 python tools/iisph_time.py --dims 256 256 256 --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  iisph', round(d['ms_per_step'],3))" >> gpurun_out/s5_rl2.txt || exit 1
  FSA_XCD_RUN_LOG=$r timeout -k 10 300 python tools/dfsph_time.py --dims 256 256 256 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  dfsph', round(d['ms_per_step'],3))" >> gpurun_out/s5_rl2.txt || exit 1
  FSA_XCD_RUN_LOG=$r timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  wcsph', round(d['ms_per_step'],4))" >> gpurun_out/s5_rl2.txt || exit 1
done
