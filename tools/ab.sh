# A/B of one environment switch on the training step:  bash tools/ab.sh VAR "v1 v2 ..." [repeats] [log name]
mkdir -p gpurun_out/ab
LOG=gpurun_out/ab/${4:-train}.log
for r in $(seq 1 ${3:-2}); do
for v in $2; do
  echo -n "$1=$v  " >> $LOG
  env $1=$v timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])" >> $LOG
done; done
cat $LOG
