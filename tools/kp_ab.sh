set -x
mkdir -p gpurun_out/kp
timeout 400 python -m pytest tests/test_ipa_pair.py tests/test_network.py -x -q -m gpu > gpurun_out/kp/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/kp/tests.log
tail -3 gpurun_out/kp/tests.log
for s in "30 128" "1 128" "8 256"; do timeout 120 python tools/bench_ipa_attn.py $s 2>&1 | grep "B="; done | tee gpurun_out/kp/micro.log
for r in 1 2; do
for v in 0 1; do
  echo "FD_IPA_KP_SOA=$v" >> gpurun_out/kp/train.log
  FD_IPA_KP_SOA=$v timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])" >> gpurun_out/kp/train.log
done; done
cat gpurun_out/kp/train.log
for v in 0 1; do
  echo "sample FD_IPA_KP_SOA=$v" >> gpurun_out/kp/sample.log
  FD_IPA_KP_SOA=$v timeout 200 python bench.py --mode sample --n-res 128 --batch 1 --num-t 100 --steps 1 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['unit'])" >> gpurun_out/kp/sample.log
done
cat gpurun_out/kp/sample.log
