"""Where a wave of the fused edge-transition kernel spends its cycles: builds a second copy of the library with
csrc/fd_edge_mlp.hip compiled under -DEM_PHASE_TIMING (s_memtime at stage boundaries, summed over waves) and runs the training
variants at B=30 x N=128.  Probe only: the product library has no timing code.

    python tools/probes/edge_phases.py            (on the GPU box)
"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from se3_diffusion_amd import build, hip  # noqa: E402

PHASES = ["tile prologue", "stage waits (vmcnt + barrier)", "layer-1 stages", "epilogue 1", "layer-2 stages", "epilogue 2",
          "layer-3 stages", "final epilogue (backward)", "final epilogue + layer 4 (forward)", "fused bwd prologue"]


def build_probe(extra=(), tag="phases"):
    build.build(verbose=False)
    out = os.path.join(ROOT, "tools", "probes", f"libfd_{tag}.so")
    obj = os.path.join(ROOT, "tools", "probes", f"fd_edge_mlp_{tag}.o")
    src = os.path.join(build.CSRC, "fd_edge_mlp.hip")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        flags = list(extra) if extra else ["-DEM_PHASE_TIMING"]
        subprocess.check_call([build.HIPCC, *build.FLAGS, "-DFD_PROBE_BUILD", *flags, "-c", src, "-o", obj])
        objs = [os.path.join(build.OBJ, f) for f in sorted(os.listdir(build.OBJ)) if f.endswith(".o") and f != "fd_edge_mlp.o"]
        subprocess.check_call([build.HIPCC, f"--offload-arch={build.ARCH}", "-shared", "-fPIC", obj, *objs, "-o", out])
    return out


def main():
    if "--plain" in sys.argv:
        L = ctypes.CDLL(hip.LIB_PATH)          # the product library, same harness: the reference time of the ablations
    elif "--ablate-pq" in sys.argv:
        # the same run with the per-residue terms P1_i / Q1_j / Pf_i / Qf_j replaced by constants: what their fetches cost
        L = ctypes.CDLL(build_probe(("-DEM_ABLATE_PQ",), "ablate_pq"))
    else:
        L = ctypes.CDLL(build_probe())
    dev = "cuda"
    # (--b / --n: the launch's size; below 65,536 pair rows the 4-wave shape -- the instrumented source -- runs, e.g. --b 1: a lone backbone)
    B = int(sys.argv[sys.argv.index("--b") + 1]) if "--b" in sys.argv else 30
    N = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 128
    R, P = B * N, B * N * N
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
    e = lambda *s: torch.empty(*s, device=dev)
    W1, W2, Wf, W40 = rn(384, 384, sc=0.08), rn(384, 384, sc=0.08), rn(128, 384, sc=0.08), rn(40, 128, sc=0.1)
    z, P1, Q1, Pf, Qf = rn(P, 128), rn(R, 384, sc=0.5), rn(R, 384, sc=0.5), rn(R, 128, sc=0.5), rn(R, 128, sc=0.5)
    b2, gm, bt, b40, emask = rn(384, sc=0.3), 1 + rn(128, sc=0.2), rn(128, sc=0.2), rn(40), torch.ones(P, device=dev)
    img = torch.empty(hip.EDGE_MLP_IMAGE_BYTES, dtype=torch.uint8, device=dev)
    imgT, imgB = torch.empty_like(img), torch.empty_like(img)
    # the product's own packers give the exact images: use them through the product binding, then time with the probe library
    from se3_diffusion_amd import ops
    img = ops.edge_mlp_pack(W1, W2, Wf, W40=W40)
    imgT = ops.edge_mlp_pack(W1, W2, Wf, backward=True)
    imgB = ops.edge_mlp_pack_bwd(Wf, W2, W1, W40=W40)
    out, h1, h2, y, mean, rstd, zb = e(P, 128), e(P, 384), e(P, 384), e(P, 128), e(P), e(P), e(P, 40)
    mh1 = torch.zeros(P, 12, dtype=torch.int32, device=dev); mh2 = torch.zeros_like(mh1)
    dz, d2, d1, dy, dzb = e(P, 128), e(P, 384), e(P, 384), e(P, 128), rn(P, 40)
    dg, db = torch.zeros(128, device=dev), torch.zeros(128, device=dev)
    up = rn(P, 128)

    def desc(**kw):
        d = hip.FdEdgeMlpDesc()
        for k, v in kw.items():
            setattr(d, k, v.data_ptr() if torch.is_tensor(v) else v)
        d.rows, d.nres, d.eps = P, N, 1e-5
        return d

    cases = {
        "forward, inference (no saves)": desc(x=z, img=img, out=out, p1=P1, q1=Q1, bias2=b2, pf=Pf, qf=Qf, gamma=gm, beta=bt,
                                               rowscale=emask),
        "forward, inference + zb (sampling)": desc(x=z, img=img, out=out, p1=P1, q1=Q1, bias2=b2, pf=Pf, qf=Qf, gamma=gm, beta=bt,
                                                    rowscale=emask, zb_out=zb, zb_bias=b40),
        "forward, training (saves + zb)": desc(x=z, img=img, out=out, p1=P1, q1=Q1, bias2=b2, pf=Pf, qf=Qf, gamma=gm, beta=bt,
                                               rowscale=emask, save1=h1, save2=h2, y=y, mean=mean, rstd=rstd, zb_out=zb,
                                               zb_bias=b40, mask1=mh1, mask2=mh2),
        "backward (packed gates, saves)": desc(x=y, img=imgT, out=dz, gmask1=mh2, gmask2=mh1, save1=d2, save2=d1, backward=1),
        "backward, fused LN + dzb prologue": desc(x=up, img=imgB, out=dz, gmask1=mh2, gmask2=mh1, save1=d2, save2=d1, backward=1,
                                                  ln_y=y, ln_mean=mean, ln_rstd=rstd, ln_gamma=gm, ln_rowscale=emask, dy_out=dy,
                                                  ln_dgamma=dg, ln_dbeta=db, dzb=dzb),
    }
    L.fd_edge_mlp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    buf = (ctypes.c_ulonglong * 12)()
    for name, d in cases.items():
        for _ in range(2):
            assert L.fd_edge_mlp(ctypes.byref(d), None) == 0
        torch.cuda.synchronize()
        if hasattr(L, "fd_edge_mlp_phases"):
            L.fd_edge_mlp_phases(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            assert L.fd_edge_mlp(ctypes.byref(d), None) == 0
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        if not hasattr(L, "fd_edge_mlp_phases"):
            print(f"{name}: {ms:.3f} ms per launch (no timers)")
            continue
        L.fd_edge_mlp_phases(buf, 0)
        tot = float(sum(buf[i] for i in range(10)))
        blocks = buf[10] / reps
        print(f"\n{name}: {ms:.3f} ms per launch (instrumented), {blocks:.0f} blocks; wave cycles by phase "
              f"(s_memtime ticks, all waves = 100 %):")
        for i, ph in enumerate(PHASES):
            if buf[i]:
                print(f"  {ph:32s} {100.0 * buf[i] / tot:5.1f} %   {buf[i] / reps / (blocks * 4):12.0f} ticks per wave and launch")
        print(f"  total {tot / reps / (blocks * 4):.0f} ticks per wave and launch; pure MFMA issue = 3072 MFMAs x 16 cycles per tile "
              f"x {P / 16 / (blocks * 4):.1f} tiles per wave")


if __name__ == "__main__":
    main()
