python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" > gpurun_out/r03_k_prio.log 2>&1
for cfg in "FD_SIDE_PRIORITY=0 FD_NONE=1" "FD_SIDE_PRIORITY=1 FD_NONE=1" "FD_SIDE_PRIORITY=0 FD_BENCH_MAIN_PRIORITY=-1" "FD_SIDE_PRIORITY=1 FD_BENCH_MAIN_PRIORITY=-1" "FD_SIDE_PRIORITY=0 FD_NONE=1" "FD_SIDE_PRIORITY=1 FD_BENCH_MAIN_PRIORITY=-1"; do
  echo -n "$cfg  " >> gpurun_out/r03_k_prio.log
  env $cfg timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])" >> gpurun_out/r03_k_prio.log
done
cat gpurun_out/r03_k_prio.log
