# Everything the round's DESIGN.md / profiles/ quote, from ONE box (box-to-box spread is ~8 %):
#   bench line, per-kernel time of the step (rocprofv3 --kernel-trace --stats), PMC of the fused edge kernel,
#   whole-step HBM traffic, full 500-step sampling runs.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05}
rm -rf $O; mkdir -p $O
timeout 120 python -m pytest tests/test_switches.py -x -q -m gpu > $O/test_switches.log 2>&1
python bench.py > $O/bench_train.json 2> $O/bench_train.err
python bench.py --mixed-n --steps 12 --warmup 3 > $O/bench_mixed_n.json 2> $O/bench_mixed_n.err
python bench.py --mode sample --n-res 128 --batch 1 --steps 1 --warmup 1 > $O/bench_sample_n128_b1.json 2>/dev/null
python bench.py --mode sample --n-res 256 --batch 1 --steps 1 --warmup 1 > $O/bench_sample_n256_b1.json 2>/dev/null
python bench.py --mode sample --n-res 512 --batch 8 --steps 1 --warmup 0 > $O/bench_sample_n512_b8.json 2>/dev/null
python bench.py --mode sample --n-res 128 --batch 32 --steps 1 --warmup 0 > $O/bench_sample_n128_b32.json 2>/dev/null
python tools/bench_edge_mlp.py --shapes 30x128,8x128,1x128,1x256,1x512 > $O/edge_mlp_microbench.log 2>&1
python tools/bench_pair_dw.py > $O/pair_dw_microbench.log 2>&1
python tools/bench_group_dw.py > $O/group_dw_microbench.log 2>&1
python tools/bench_embed_bwd.py > $O/embed_bwd_microbench.log 2>&1
python tools/bench_ipa_attn.py 30 128 > $O/ipa_attn_microbench.log 2>&1
(timeout 200 python tools/bench_ipa_flash.py; timeout 200 python tools/bench_ipa_flash.py --bwd) > $O/ipa_flash_microbench.log 2>&1
python tools/bench_gemm.py --only kk --iters 50 > $O/gemm_s64_microbench.log 2>&1
python tools/bench_node_gemm.py 3840 > $O/node_gemm.log 2>&1
python tools/gemm_shapes.py 30 128 > $O/gemm_shapes.log 2>&1
python tools/bench_ipa_keys.py > $O/ipa_keys_microbench.log 2>&1
# per-kernel time of the training step, launches serialised
FD_BENCH_PROFILE=1 FD_GRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python bench.py --steps 5 --warmup 2 --no-sampling --no-cpu-baseline > $O/kt.log 2>&1
python tools/kernel_stats_md.py $O/kt/p_kernel_stats.csv "training step B=30 x N=128, FD_GRAD_STREAM=0 (serialised): 4 priming + 2 warm-up + 5 timed + 3 profiled steps of bench.py" > $O/train_kernel_stats.md
# sampling forward kernels
rocprofv3 --kernel-trace --stats -d $O/ks -o p --output-format csv -- python bench.py --mode sample --n-res 128 --batch 1 --num-t 100 --steps 1 --warmup 0 --no-graph > $O/ks.log 2>&1
python tools/kernel_stats_md.py $O/ks/p_kernel_stats.csv "sampling N=128 B=1, 100 steps, eager launches" > $O/sample_n128_b1_kernel_stats.md
rocprofv3 --kernel-trace --stats -d $O/ks2 -o p --output-format csv -- python bench.py --mode sample --n-res 256 --batch 1 --num-t 100 --steps 1 --warmup 0 --no-graph > $O/ks2.log 2>&1
python tools/kernel_stats_md.py $O/ks2/p_kernel_stats.csv "sampling N=256 B=1, 100 steps, eager launches" > $O/sample_n256_b1_kernel_stats.md
rocprofv3 --kernel-trace --stats -d $O/ks5 -o p --output-format csv -- python bench.py --mode sample --n-res 512 --batch 8 --num-t 12 --steps 1 --warmup 0 --no-graph > $O/ks5.log 2>&1
python tools/kernel_stats_md.py $O/ks5/p_kernel_stats.csv "sampling N=512 B=8, 12 steps, eager launches" > $O/sample_n512_b8_kernel_stats.md
if [ -z "$LITE" ]; then   # (LITE=1: the counter passes are skipped -- kernels unchanged since the last full run)
bash tools/pmc_roofline.sh ${1:-r05} > $O/pmc_roofline.txt 2>&1      # HBM bytes per kernel of the step, calibrated -> bench.py
bash tools/pmc_step_sq.sh > $O/pmc_step_sq.txt 2>&1                   # SQ / LDS / clock counters of the big kernels IN the step
fi
bash tools/prof_gap.sh > $O/step_gap.txt 2>&1
find $O -name "*.csv" -size +512k -delete
ls -la $O
