# SQ / LDS / clock counters of the dominant kernels as the TRAINING STEP runs them (forward with saves + zb layer, backward,
# grouped weight gradients), gradient side stream off so every dispatch has the chip to itself.  Two passes (8 SQ slots each,
# MI355X_MICROARCH.md "rocprofv3 PMC slots"), counters in their own runs with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_step_sq
rm -rf $O; mkdir -p $O
CMD="python bench.py --steps 2 --warmup 1 --no-sampling --no-cpu-baseline"
FD_BENCH_PROFILE=1 FD_BENCH_PRIME=1 FD_GRAD_STREAM=0 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/p1 -o p --output-format csv -- $CMD > $O/p1.log 2>&1
FD_BENCH_PROFILE=1 FD_BENCH_PRIME=1 FD_GRAD_STREAM=0 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/p2 -o p --output-format csv -- $CMD > $O/p2.log 2>&1
for k in edge_mlp16_kernel pair_dw_kernel group_dw_kernel edge_embed ipa_flash gemm_w_kernel seq_attn_bwd; do
  echo "==== $k"
  python tools/pmc_summary.py $O/p1 $k
  python tools/pmc_summary.py $O/p2 $k
done
find $O -name "*.csv" -size +1M -delete
