# A/B of one environment switch on sampling:  bash tools/ab_sample.sh VAR "v1 v2 ..." N B [num_t] [repeats]
mkdir -p gpurun_out/ab
LOG=gpurun_out/ab/sample_$1_n$3_b$4.log
for r in $(seq 1 ${6:-2}); do
for v in $2; do
  echo -n "$1=$v N=$3 B=$4  " >> $LOG
  env $1=$v timeout 300 python bench.py --mode sample --n-res $3 --batch $4 --num-t ${5:-500} --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d.get('ms_per_step'), d.get('value'))" >> $LOG
done; done
cat $LOG
