# GPU busy vs wall per training step (kernel trace with the side stream on, as shipped)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_gap
rm -rf $O; mkdir -p $O
FD_BENCH_PROFILE=1 rocprofv3 --kernel-trace -d $O -o p --output-format csv -- python bench.py --steps 6 --warmup 2 --no-sampling --no-cpu-baseline > $O/log.txt 2>&1
python tools/step_gap.py $(find $O -name "*kernel_trace.csv") $O/timeline.txt > $O/gap.txt 2>&1
find $O -name "*.csv" -size +1M -delete
cat $O/gap.txt | head -70
