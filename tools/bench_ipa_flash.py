"""fd_ipa_flash_fwd against the launch sequence it replaces (q k^T GEMM -> fd_ipa_attn_fwd -> a v / a v_pts GEMMs ->
fd_ipa_opt_fwd) at the shapes of the benchmark configs: us per IPA block, HIP events over back-to-back replays.
    python tools/bench_ipa_flash.py [B N [hpb ...]]      (GPU box)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_ipa_flash as T  # noqa: E402  (input builder + the replaced sequence: test infrastructure, timing only)
from se3_diffusion_amd import ops  # noqa: E402


_FLUSH = None


def timeit(fn, reps=20, warm=3):
    """us per call.  --cold: every call is preceded (outside the timed window) by a 1 GB write that pushes the operands out
    of the 256 MB Infinity Cache -- in the step zb arrives from HBM: the edge transition that writes it moves 2.5 GB at N = 512."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if "--cold" in sys.argv:
        global _FLUSH
        if _FLUSH is None:
            _FLUSH = torch.empty(256 << 20, device="cuda")
        tot = 0.0
        for _ in range(8):
            _FLUSH.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / 8 * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def case(B, N, hpbs, spread=1.0):
    L = ops.lib()
    dev = "cuda"
    proj, quat, trans, zb, hw, mask = T._inputs(dev, B, N, 0, spread, masked=False)
    qp, kp, vp, kpT = T._points(L, proj, quat, trans, B, N)
    only = "--flash-only" in sys.argv
    t_seq = 0.0 if only else timeit(lambda: T._sequence(L, proj, quat, trans, zb, hw, mask, qp, kp, vp, kpT, B, N))
    f_seq, A_seq = T._sequence(L, proj, quat, trans, zb, hw, mask, qp, kp, vp, kpT, B, N)
    line = [f"B={B:3d} N={N:4d}: sequence (5 launches) {t_seq:8.1f} us"]
    feats = torch.empty(B * N, T.LDF, device=dev)
    A = torch.empty(B, T.H, N, N, device=dev)
    for hpb in hpbs:
        t = timeit(lambda: L.call("fd_ipa_flash_fwd", proj, zb, qp, kp, vp, hw, mask, quat, trans, feats, None, B, N, hpb))
        tA = timeit(lambda: L.call("fd_ipa_flash_fwd", proj, zb, qp, kp, vp, hw, mask, quat, trans, feats, A, B, N, hpb))
        err = float((feats - f_seq).abs().max() / f_seq.abs().max())
        errA = float((A - A_seq).abs().max())
        line.append(f"flash hpb={hpb}: {t:8.1f} us (with A {tA:8.1f}) maxdiff {err:.1e} / A {errA:.1e}")
    print(" | ".join(line), flush=True)


# compile-time variants of the kernel (prefetch distances, scheduling pins): second copies of the library with
# fd_ipa_flash.hip recompiled under -D flags, built in the build container (`--build`) so that they travel with gpurun
VARIANTS = {"abl_nodzb": ["-DFL_ABL_NODZB"], "abl_nopts": ["-DFL_ABL_NOPTS"], "abl_nodma": ["-DFL_ABL_NODMA"], "abl_nokv": ["-DFL_ABL_NOKV"],
            "abl_nobar": ["-DFL_ABL_NOBAR"], "abl_all": ["-DFL_ABL_NODMA", "-DFL_ABL_NOKV", "-DFL_ABL_NODZB", "-DFL_ABL_NOPTS"]}
PROBES = os.path.join(ROOT, "tools", "probes")


def build_variants():
    import subprocess
    from se3_diffusion_amd import build
    build.build(verbose=False)
    others = [os.path.join(build.OBJ, f) for f in sorted(os.listdir(build.OBJ)) if f.endswith(".o") and f != "fd_ipa_flash.o"]
    for tag, flags in VARIANTS.items():
        obj = os.path.join(PROBES, f"fd_ipa_flash_{tag}.o")
        r = subprocess.run([build.HIPCC, *build.FLAGS, "-DFD_PROBE_BUILD", *flags, "-c", os.path.join(build.CSRC, "fd_ipa_flash.hip"), "-o", obj,
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        spills = [l.split("VGPRs Spill:")[1].split()[0] for l in r.stderr.splitlines() if "VGPRs Spill:" in l]
        subprocess.check_call([build.HIPCC, f"--offload-arch={build.ARCH}", "-shared", "-fPIC", obj, *others, "-o",
                               os.path.join(PROBES, f"libfd_flash_{tag}.so")])
        os.remove(obj)
        print(f"{tag}: built, spilled VGPRs per kernel {spills}")


def variants(B, N, hpb, spread=1.0):
    from se3_diffusion_amd import hip
    dev = "cuda"
    L0 = ops.lib()
    proj, quat, trans, zb, hw, mask = T._inputs(dev, B, N, 0, spread, masked=False)
    qp, kp, vp, kpT = T._points(L0, proj, quat, trans, B, N)
    feats = torch.empty(B * N, T.LDF, device=dev)
    f0 = torch.empty_like(feats)
    L0.call("fd_ipa_flash_fwd", proj, zb, qp, kp, vp, hw, mask, quat, trans, f0, None, B, N, hpb)
    line = [f"B={B:3d} N={N:4d} hpb={hpb}"]
    for tag in ["shipped"] + sorted(VARIANTS):
        path = hip.LIB_PATH if tag == "shipped" else os.path.join(PROBES, f"libfd_flash_{tag}.so")
        if not os.path.exists(path):
            continue
        L = hip.FdLib(path)
        t = timeit(lambda: L.call("fd_ipa_flash_fwd", proj, zb, qp, kp, vp, hw, mask, quat, trans, feats, None, B, N, hpb))
        line.append(f"{tag} {t:7.1f} us{'' if torch.equal(feats, f0) else ' (DIFFERS)'}")
    print(" | ".join(line), flush=True)


def case_split(B, N, spread=1.0):
    """a lone backbone: the launch sequence against the key-split kernel + merge launch (fd_ipa_flash_fwd_split)"""
    L = ops.lib()
    dev = "cuda"
    proj, quat, trans, zb, hw, mask = T._inputs(dev, B, N, 0, spread, masked=False)
    qp, kp, vp, kpT = T._points(L, proj, quat, trans, B, N)
    t_seq = timeit(lambda: T._sequence(L, proj, quat, trans, zb, hw, mask, qp, kp, vp, kpT, B, N))
    f_seq, _ = T._sequence(L, proj, quat, trans, zb, hw, mask, qp, kp, vp, kpT, B, N)
    feats = torch.empty(B * N, T.LDF, device=dev)
    line = [f"B={B:3d} N={N:4d}: sequence (5 launches) {t_seq:7.1f} us"]
    for hpb in (4, 2):
        for ks in (1, 2, 4, 8):
            part = torch.empty(max(1, ks * B * N * T.H * 328), device=dev)
            t = timeit(lambda: L.call("fd_ipa_flash_fwd_split", proj, zb, qp, kp, vp, hw, mask, quat, trans, feats, None, B, N, hpb,
                                      ks, part))
            err = float((feats - f_seq).abs().max() / f_seq.abs().max())
            line.append(f"hpb={hpb} x{ks}: {t:6.1f} ({err:.0e})")
    print(" | ".join(line), flush=True)


def case_bwd(B, N, spread=1.0):
    """the query side of the backward: dA GEMM + fd_ipa_opt_bwd + dA += GEMM + fd_ipa_attn_bwd against fd_ipa_opt_bwd_dot +
    fd_ipa_flash_bwd (both include the head-weight column sum and fd_ipa_kpts_bwd)"""
    L = ops.lib()
    dev = "cuda"
    proj, quat, trans, zb, hw, mask = T._inputs(dev, B, N, 0, spread, masked=False)
    qp, kp, vp, kpT = T._points(L, proj, quat, trans, B, N)
    feats, A = T._sequence(L, proj, quat, trans, zb, hw, mask, qp, kp, vp, kpT, B, N)
    dfeats = torch.randn(B * N, T.LDF, device=dev)
    t_seq = timeit(lambda: T._bwd_sequence(L, proj, quat, zb, hw, qp, kp, vp, kpT, A, feats, dfeats, B, N))
    t_fl = timeit(lambda: T._bwd_flash(L, proj, quat, trans, zb, hw, qp, kp, vp, A, feats, dfeats, B, N, poison=False))
    ref = T._bwd_sequence(L, proj, quat, zb, hw, qp, kp, vp, kpT, A, feats, dfeats, B, N)
    out = T._bwd_flash(L, proj, quat, trans, zb, hw, qp, kp, vp, A, feats, dfeats, B, N)
    err = {k: float((out[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30)) for k in ref}
    print(f"B={B:3d} N={N:4d} backward, query side: sequence {t_seq:8.1f} us | flash {t_fl:8.1f} us | maxdiff "
          + " ".join(f"{k} {v:.1e}" for k, v in err.items()), flush=True)


def variants_bwd(B, N, spread=1.0):
    from se3_diffusion_amd import hip
    dev = "cuda"
    L0 = ops.lib()
    proj, quat, trans, zb, hw, mask = T._inputs(dev, B, N, 0, spread, masked=False)
    qp, kp, vp, kpT = T._points(L0, proj, quat, trans, B, N)
    feats, A = T._sequence(L0, proj, quat, trans, zb, hw, mask, qp, kp, vp, kpT, B, N)
    dfeats = torch.randn(B * N, T.LDF, device=dev)
    R = B * N
    doptg = torch.empty(R, T.H, T.PV * 3, device=dev); dframe = torch.zeros(R, 12, device=dev); ptdot = torch.empty(R, T.H, device=dev)
    L0.call("fd_ipa_opt_bwd_dot", dfeats, feats, quat, trans, doptg, dframe, ptdot, R)
    dL = torch.empty(B, T.H, N, N, device=dev); dzb = torch.empty(R * N, T.ZB, device=dev)
    dqp = torch.empty(R, T.H, 24, device=dev); dkp = torch.empty(R, T.H, 24, device=dev)
    dhw = torch.zeros(T.H, device=dev); part = torch.empty(R, T.H, device=dev)
    line = [f"B={B:3d} N={N:4d} fd_ipa_flash_bwd (incl. column sum + fd_ipa_kpts_bwd)"]
    for tag in ["shipped"] + sorted(VARIANTS):
        path = hip.LIB_PATH if tag == "shipped" else os.path.join(PROBES, f"libfd_flash_{tag}.so")
        if not os.path.exists(path):
            continue
        L = hip.FdLib(path)
        t = timeit(lambda: L.call("fd_ipa_flash_bwd", proj, A, zb, dfeats, feats, doptg, ptdot, qp, kp, vp, hw, trans, dL, dzb, dqp,
                                  dkp, dhw, part, B, N))
        line.append(f"{tag} {t:7.1f} us")
    t = timeit(lambda: L0.call("fd_ipa_kpts_bwd", dL, qp, kp, hw, dkp, B, N))
    line.append(f"(fd_ipa_kpts_bwd alone {t:6.1f} us)")
    print(" | ".join(line), flush=True)


def main():
    if "--split" in sys.argv:
        for (B, N, sp) in ((1, 128, 1.0), (1, 256, 1.2), (1, 512, 1.5), (2, 256, 1.2), (4, 128, 1.0)):
            case_split(B, N, sp)
        return
    if "--variants-bwd" in sys.argv:
        variants_bwd(30, 128)
        variants_bwd(8, 512, spread=1.5)
        return
    if "--bwd" in sys.argv:
        case_bwd(30, 128)
        case_bwd(8, 512, spread=1.5)
        case_bwd(7, 256, spread=1.2)
        return
    if "--build" in sys.argv:
        build_variants()
        return
    if "--variants" in sys.argv:
        variants(30, 128, 8)
        variants(8, 512, 8, spread=1.5)
        variants(7, 256, 4, spread=1.2)
        return
    argv = [x for x in sys.argv[1:] if not x.startswith("--")]
    if len(argv) >= 2:
        case(int(argv[0]), int(argv[1]), [int(x) for x in argv[2:]] or [0])
        return
    case(30, 128, [8, 4])
    case(8, 512, [8], spread=1.5)
    case(7, 256, [8, 4], spread=1.2)
    case(1, 128, [2, 4, 8])
    case(1, 256, [2, 4, 8], spread=1.2)
    case(1, 512, [4, 8], spread=1.5)


if __name__ == "__main__":
    main()
