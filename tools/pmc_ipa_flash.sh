# PMC passes for fd_ipa_flash_fwd (separate --pmc passes, kernel-trace only; MI355X_MICROARCH.md)   bash tools/pmc_ipa_flash.sh [B N hpb]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_ipa_flash
mkdir -p $O
CMD="python tools/bench_ipa_flash.py --flash-only ${1:-8} ${2:-512} ${3:-8}"     # (flash launches only: a --pmc pass serialises every kernel)
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/p1 -o p1 --output-format csv -- $CMD > $O/p1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/p2 -o p2 --output-format csv -- $CMD > $O/p2.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d $O/p3 -o p3 --output-format csv -- $CMD > $O/p3.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum -d $O/p4 -o p4 --output-format csv -- $CMD > $O/p4.log 2>&1
for p in p1 p2 p3 p4; do python tools/pmc_summary.py $O/$p ipa_flash > $O/$p.summary 2>&1; done
find $O -name "*.csv" -size +1M -delete
cat $O/*.summary; tail -3 $O/p4.log
