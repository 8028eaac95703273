#!/bin/bash
# round 5, GPU call E: per-kernel time of the training step (serialised) and of the sampling forward on the current code
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e
rm -rf $O; mkdir -p $O
FD_BENCH_PROFILE=1 FD_GRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python bench.py --steps 5 --warmup 2 --no-sampling --no-cpu-baseline > $O/kt.log 2>&1
python tools/kernel_stats_md.py $O/kt/p_kernel_stats.csv "training step B=30 x N=128, FD_GRAD_STREAM=0 (serialised): 4 priming + 2 warm-up + 5 timed + 3 profiled steps of bench.py" > $O/train_kernel_stats.md
rocprofv3 --kernel-trace --stats -d $O/ks -o p --output-format csv -- python bench.py --mode sample --n-res 128 --batch 1 --num-t 100 --steps 1 --warmup 0 --no-graph > $O/ks.log 2>&1
python tools/kernel_stats_md.py $O/ks/p_kernel_stats.csv "sampling N=128 B=1, 100 steps, eager launches" > $O/sample_n128_b1_kernel_stats.md
find $O -name "*.csv" -size +512k -delete
head -45 $O/train_kernel_stats.md
head -30 $O/sample_n128_b1_kernel_stats.md
