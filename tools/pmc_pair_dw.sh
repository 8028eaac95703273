# PMC passes for the grouped pair-row weight-gradient kernel (separate --pmc passes, kernel-trace only; MI355X_MICROARCH.md)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_pair_dw
mkdir -p $O
CMD="python tools/bench_pair_dw.py"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/p1 -o p1 --output-format csv -- $CMD > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d $O/p2 -o p2 --output-format csv -- $CMD > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p3 -o p3 --output-format csv -- $CMD > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/p4 -o p4 --output-format csv -- $CMD > $O/p4.log 2>&1
for p in p1 p2 p3 p4; do python tools/pmc_summary.py $O/$p pair_dw > $O/$p.summary 2>&1; done
find $O -name "*.csv" -size +1M -delete
cat $O/*.summary
