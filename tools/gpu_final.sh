#!/bin/bash
# final validation of a round on one box: the GPU test tier (with the achieved parity errors recorded), smoke(), the default bench line
O=gpurun_out/final
rm -rf $O gpurun_out/parity_errors.jsonl; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -5 $O/gputest.log
grep "\[parity\]" $O/gputest.log | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final/bench_default.json').read())
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_detail']['source'])
print({k:(v['backbones_per_s'] if isinstance(v,dict) else None) for k,v in d['config']['sampling'].items()})
print(d['config']['mixed_n']['residues_per_s'], d['cpu_baseline']['value'])
PY
