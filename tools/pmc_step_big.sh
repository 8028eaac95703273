# PMC of the three big kernels INSIDE the training step (real operands: ReLU-gated activations and gradients)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_step_big
rm -rf $O; mkdir -p $O
CMD="python bench.py --steps 4 --warmup 2 --no-sampling --no-cpu-baseline"
FD_BENCH_PROFILE=1 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/p1 -o p1 --output-format csv -- $CMD > $O/p1.log 2>&1
for k in pair_dw edge_mlp16; do python tools/pmc_summary.py $O/p1 $k; done > $O/summary.txt 2>&1
find $O -name "*.csv" -size +1M -delete
cat $O/summary.txt
