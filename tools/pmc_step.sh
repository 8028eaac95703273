# whole-step HBM traffic of the training step (B=30 x N=128): FETCH_SIZE and WRITE_SIZE in separate --pmc passes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_step
mkdir -p $O
CMD="python bench.py --steps 3 --warmup 1 --no-sampling --no-cpu-baseline"
FD_BENCH_PROFILE=1 FD_GRAD_STREAM=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- $CMD > $O/f.log 2>&1
FD_BENCH_PROFILE=1 FD_GRAD_STREAM=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- $CMD > $O/w.log 2>&1
python tools/pmc_step_traffic.py $O/f $O/w > $O/summary.txt 2>&1
find $O -name "*.csv" -size +1M -delete
cat $O/summary.txt
