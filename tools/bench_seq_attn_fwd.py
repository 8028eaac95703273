"""fd_seq_attn_fwd at the benchmark shapes, us per launch.   python tools/bench_seq_attn_fwd.py   (GPU box)"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3_diffusion_amd.ops import lib
def timeit(fn, reps=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for B, N in ((30,128),(8,512),(1,256),(1,512)):
    qkv = torch.randn(B*N, 960, device="cuda"); out = torch.empty(B*N, 320, device="cuda"); A = torch.empty(B,4,N,N, device="cuda")
    t0 = timeit(lambda: lib().call("fd_seq_attn_fwd", qkv, None, out, None, 1/math.sqrt(80), B, N))
    t1 = timeit(lambda: lib().call("fd_seq_attn_fwd", qkv, None, out, A, 1/math.sqrt(80), B, N))
    print(f"B={B} N={N}: seq_attn_fwd {t0:.1f} us (with A {t1:.1f})")
