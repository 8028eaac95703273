# One-box ablation of the round's switches: the training step with each optimisation turned off in turn (everything else on).
cd $GRAFT_REPO_ROOT
O=gpurun_out/ablation.txt
: > $O
run() { env $1 python bench.py --no-sampling --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %7.3f ms  %9.1f residues/s' % ('$2', d['ms_per_step'], d['value']))" >> $O; }
run "FD_NONE=1" "all on (shipped)"
run "FD_PAIR_DW_BLOCKS=0" "pair_dw on all 256 CUs"
run "FD_PAIR_DW=0" "pair-row dW through fd_gemm"
run "FD_GEMM_NO_S64=1" "no 64x64 split-bf16 tile"
run "FD_ZERO_ARENA=0" "no zero arena"
run "FD_DX_SPLITK=0" "no split-K accumulating dX"
run "FD_IPA_ATTN_FUSED=0" "softmax / o_pair as two launches"
run "FD_EMBED_FUSED=0" "edge embedder unfused"
run "FD_EDGE_FUSED=0" "edge transition unfused"
run "FD_GRAD_STREAM=0" "no gradient side stream"
run "FD_GEMM_EXACT_F32=1" "every GEMM bitwise fp32"
run "FD_NONE=1" "all on (shipped), again"
cat $O
