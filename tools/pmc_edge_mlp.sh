cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/pmc_em
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d gpurun_out/pmc_em/p1 -o p1 --output-format csv -- python tools/bench_edge_mlp.py --shapes 30x128 > gpurun_out/pmc_em/p1.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/pmc_em/p2 -o p2 --output-format csv -- python tools/bench_edge_mlp.py --shapes 30x128 > gpurun_out/pmc_em/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_em/p3 -o p3 --output-format csv -- python tools/bench_edge_mlp.py --shapes 30x128 > gpurun_out/pmc_em/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc_em/p4 -o p4 --output-format csv -- python tools/bench_edge_mlp.py --shapes 30x128 > gpurun_out/pmc_em/p4.log 2>&1
for p in p1 p2 p3 p4; do python tools/pmc_summary.py gpurun_out/pmc_em/$p edge_mlp_kernel > gpurun_out/pmc_em/$p.summary 2>&1; done
find gpurun_out/pmc_em -name "*.csv" -size +2M -delete
cat gpurun_out/pmc_em/*.summary
