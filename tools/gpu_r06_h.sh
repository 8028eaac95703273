#!/bin/bash
# round 6, GPU call H: the side stream's CU shares re-swept on the round-6 kernels (same box)
O=gpurun_out/r06h
mkdir -p $O
run() { env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2; do
for v in 128 144 160 176 192 224; do echo "FD_PAIR_DW_BLOCKS=$v $(run FD_PAIR_DW_BLOCKS=$v)" >> $O/sweep.log; done
for v in 0 384 256; do echo "FD_NODE_DW_BLOCKS=$v $(run FD_NODE_DW_BLOCKS=$v)" >> $O/sweep.log; done
echo "FD_DEFER_NODE_DW=0 $(run FD_DEFER_NODE_DW=0)" >> $O/sweep.log
echo "FD_GRAD_STREAM=0 $(run FD_GRAD_STREAM=0)" >> $O/sweep.log
done
cat $O/sweep.log
