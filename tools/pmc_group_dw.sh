# PMC passes for fd_group_dw alone on one trunk block's item list (separate --pmc passes, kernel-trace only; MI355X_MICROARCH.md)
#   bash tools/pmc_group_dw.sh <tag> [FD_GROUP_DW_LOCKSTEP]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_group_dw_$1
mkdir -p $O
export FD_GROUP_DW_LOCKSTEP=${2:-1}
CMD="python tools/bench_group_dw.py 3840 0"
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p1 -o p1 --output-format csv -- $CMD > $O/p1.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p2 -o p2 --output-format csv -- $CMD > $O/p2.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/p3 -o p3 --output-format csv -- $CMD > $O/p3.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM TA_BUSY_avr -d $O/p4 -o p4 --output-format csv -- $CMD > $O/p4.log 2>&1
for p in p1 p2 p3 p4; do python tools/pmc_summary.py $O/$p group_dw > $O/$p.summary 2>&1; done
find $O -name "*.csv" -size +1M -delete
cat $O/*.summary
