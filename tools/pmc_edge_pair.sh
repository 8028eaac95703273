# SQ counters of the fused edge kernel on a lone backbone (16,384 pair rows): the column-split kernel (edge_mlp_pair_kernel) and the
# 4-wave shape it replaces (FD_EDGE_PAIR=0) -- separate --pmc passes, kernel-trace only (MI355X_MICROARCH.md)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_edge_pair
rm -rf $O; mkdir -p $O
CMD="python tools/bench_edge_mlp.py --shapes 1x128 --fwd-only"
for v in 1 0; do
  FD_EDGE_PAIR=$v rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/a$v -o p --output-format csv -- $CMD > $O/a$v.log 2>&1
  FD_EDGE_PAIR=$v rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $O/b$v -o p --output-format csv -- $CMD > $O/b$v.log 2>&1
  echo "== FD_EDGE_PAIR=$v"
  python tools/pmc_summary.py $O/a$v edge_mlp
  python tools/pmc_summary.py $O/b$v edge_mlp
done
find $O -name "*.csv" -size +1M -delete
