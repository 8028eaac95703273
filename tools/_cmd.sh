( time python -m pytest tests -m gpu -x -q ) 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py > gpurun_out/bench_default.json ) 2>&1 | tail -4
python -c "
import json; d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
