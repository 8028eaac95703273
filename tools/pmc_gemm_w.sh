# PMC passes for the pre-split-weight GEMM (separate --pmc passes, kernel-trace only; MI355X_MICROARCH.md)
#   bash tools/pmc_gemm_w.sh "<M N K tile fwd|dx [ks]>" <tag>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_gemm_w_$2
mkdir -p $O
CMD="python tools/bench_gemm_w_one.py $1"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/p1 -o p1 --output-format csv -- $CMD > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/p2 -o p2 --output-format csv -- $CMD > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -d $O/p3 -o p3 --output-format csv -- $CMD > $O/p3.log 2>&1
for p in p1 p2 p3; do python tools/pmc_summary.py $O/$p gemm_w > $O/$p.summary 2>&1; done
find $O -name "*.csv" -size +1M -delete
cat $O/*.summary
