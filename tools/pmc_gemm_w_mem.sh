# memory-side PMC passes for the pre-split-weight GEMM:  bash tools/pmc_gemm_w_mem.sh "<M N K tile fwd|dx [ks]>" <tag>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_gemm_w_mem_$2
mkdir -p $O
CMD="python tools/bench_gemm_w_one.py $1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/p1 -o p1 --output-format csv -- $CMD > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum -d $O/p2 -o p2 --output-format csv -- $CMD > $O/p2.log 2>&1
for p in p1 p2; do python tools/pmc_summary.py $O/$p gemm > $O/$p.summary 2>&1; done
find $O -name "*.csv" -size +1M -delete
cat $O/*.summary
