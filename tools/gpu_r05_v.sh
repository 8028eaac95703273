cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ks256
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o p --output-format csv -- python bench.py --mode sample --n-res 256 --batch 1 --num-t 100 --steps 1 --warmup 0 --no-graph > $O/log.txt 2>&1
python tools/kernel_stats_md.py $O/p_kernel_stats.csv "sampling N=256 B=1, 100 steps, eager launches" > gpurun_out/r05_sample_n256_b1_kernel_stats.md
find $O -name "*.csv" -size +512k -delete
head -32 gpurun_out/r05_sample_n256_b1_kernel_stats.md
