# per-kernel time of the training step (rocprofv3 --kernel-trace --stats), serialised launches
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_step
rm -rf $O; mkdir -p $O
FD_BENCH_PROFILE=1 FD_GRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $O -o p --output-format csv -- python bench.py --steps 5 --warmup 2 --no-sampling --no-cpu-baseline > $O/log.txt 2>&1
python tools/prof_summary.py $O > $O/summary.md 2>&1
find $O -name "*.csv" -size +1M -delete
head -45 $O/summary.md
