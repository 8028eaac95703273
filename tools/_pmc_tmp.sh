cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc2
timeout 140 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc2/p1 --output-format csv -- python $R/tools/bench_gemm.py --only "x3 edge_fwd_W2" --iters 3 --warm 1 > $R/gpurun_out/pmc2/p1.log 2>&1
echo rc1=$?
timeout 140 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $R/gpurun_out/pmc2/p2 --output-format csv -- python $R/tools/bench_gemm.py --only "x3 edge_fwd_W2" --iters 3 --warm 1 > $R/gpurun_out/pmc2/p2.log 2>&1
echo rc2=$?
cd $R
for p in p1 p2; do python tools/pmc_summary.py gpurun_out/pmc2/$p gemm_bx3 > gpurun_out/pmc2/$p.txt 2>&1; cat gpurun_out/pmc2/$p.txt; tail -3 gpurun_out/pmc2/$p.log; done
find gpurun_out/pmc2 -name "*.csv" -size +2M -delete
