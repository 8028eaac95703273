"""A/B of compile-time shapes of the fused edge-transition kernel (csrc/fd_edge_mlp.hip): waves per block, units per stage, ring
depth and copy distance of the weight stream.  Every variant is a second copy of the library with fd_edge_mlp.hip compiled under
its -D flags (built HERE, in the build container, so that the .so files travel with gpurun); on the GPU box the four launch kinds
of the training step + the inference forward are timed at B=30 x N=128 (and optionally other sizes) and every output is compared
bit for bit with the product library's (the variants change scheduling, not arithmetic).

    python tools/probes/edge_variants.py --build          (build container: compile the variant libraries)
    python tools/probes/edge_variants.py [--rows-b B --n N]   (GPU box: time them)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from se3_diffusion_amd import build, hip  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
# tag -> extra flags ("" = the product library itself)
VARIANTS = {
    "shipped": None,
    "shipped again": [],
}
if os.environ.get("EDGE_VARIANTS_EXTRA"):          # "tag:-DX=1,-DY=2;tag2:..."
    for item in os.environ["EDGE_VARIANTS_EXTRA"].split(";"):
        t, fl = item.split(":")
        VARIANTS[t] = fl.split(",")


def lib_path(tag):
    return os.path.join(HERE, f"libfd_ev_{tag}.so")


def build_all():
    build.build(verbose=False)
    # both shapes of the kernel (fd_edge_mlp.hip = 4 waves, fd_edge_mlp_w8.hip = 8 waves) are recompiled under the variant's flags
    # (EDGE_VARIANTS_SOURCES="fd_pair_dw,...": the variant's flags go to those sources instead -- step-level A/Bs of other kernels)
    names = tuple(os.environ.get("EDGE_VARIANTS_SOURCES", "fd_edge_mlp,fd_edge_mlp_w8").split(","))
    others = [os.path.join(build.OBJ, f) for f in sorted(os.listdir(build.OBJ)) if f.endswith(".o") and f[:-2] not in names]
    for tag, flags in VARIANTS.items():
        if not flags:
            continue
        objs, report = [], []
        for n in names:
            obj = os.path.join(HERE, f"{n}_ev_{tag}.o")
            r = subprocess.run([build.HIPCC, *build.FLAGS, *flags, "-c", os.path.join(build.CSRC, n + ".hip"), "-o", obj,
                                "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
            if r.returncode:
                print(tag, n, "FAILED to compile:\n", r.stderr[-1500:])
                break
            spills = [l.split("VGPRs Spill:")[1].split()[0] for l in r.stderr.splitlines() if "VGPRs Spill:" in l]
            report.append(f"{n} spills {spills[-7:]}")
            objs.append(obj)
        else:
            subprocess.check_call([build.HIPCC, f"--offload-arch={build.ARCH}", "-shared", "-fPIC", *objs, *others, "-o", lib_path(tag)])
            print(f"{tag}: built; " + "; ".join(report))
        for o in objs:
            os.remove(o)


def main():
    if "--build" in sys.argv:
        return build_all()
    import torch
    from se3_diffusion_amd import ops
    B = int(sys.argv[sys.argv.index("--rows-b") + 1]) if "--rows-b" in sys.argv else 30
    N = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 128
    dev = "cuda"
    R, P = B * N, B * N * N
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
    e = lambda *s: torch.empty(*s, device=dev)
    W1, W2, Wf, W40 = rn(384, 384, sc=0.08), rn(384, 384, sc=0.08), rn(128, 384, sc=0.08), rn(40, 128, sc=0.1)
    z, P1, Q1, Pf, Qf = rn(P, 128), rn(R, 384, sc=0.5), rn(R, 384, sc=0.5), rn(R, 128, sc=0.5), rn(R, 128, sc=0.5)
    b2, gm, bt, b40, emask = rn(384, sc=0.3), 1 + rn(128, sc=0.2), rn(128, sc=0.2), rn(40), torch.ones(P, device=dev)
    img = ops.edge_mlp_pack(W1, W2, Wf, W40=W40)
    imgB = ops.edge_mlp_pack_bwd(Wf, W2, W1, W40=W40)
    out, h1, h2, y, mean, rstd, zb = e(P, 128), e(P, 384), e(P, 384), e(P, 128), e(P), e(P), e(P, 40)
    mh1 = torch.zeros(P, 12, dtype=torch.int32, device=dev)
    mh2 = torch.zeros_like(mh1)
    dz, d2, d1, dy, dzb = e(P, 128), e(P, 384), e(P, 384), e(P, 128), rn(P, 40)
    dg, db = torch.zeros(128, device=dev), torch.zeros(128, device=dev)
    up = rn(P, 128)
    sched = torch.zeros(4, dtype=torch.int32, device=dev)
    shape = int(os.environ.get("FD_EDGE_SHAPE", "0"))

    def desc(**kw):
        d = hip.FdEdgeMlpDesc()
        for k, v in kw.items():
            setattr(d, k, v.data_ptr() if torch.is_tensor(v) else v)
        d.rows, d.nres, d.eps = P, N, 1e-5
        d.sched = sched.data_ptr()
        d.shape = shape
        return d

    cases = {
        "fwd inference": (desc(x=z, img=img, out=out, p1=P1, q1=Q1, bias2=b2, pf=Pf, qf=Qf, gamma=gm, beta=bt, rowscale=emask,
                               zb_out=zb, zb_bias=b40), (out, zb)),
        "fwd training": (desc(x=z, img=img, out=out, p1=P1, q1=Q1, bias2=b2, pf=Pf, qf=Qf, gamma=gm, beta=bt, rowscale=emask,
                              save1=h1, save2=h2, y=y, mean=mean, rstd=rstd, zb_out=zb, zb_bias=b40, mask1=mh1, mask2=mh2),
                         (out, h1, h2, y, zb, mh1, mh2)),
        "bwd fused LN+dzb": (desc(x=up, img=imgB, out=dz, gmask1=mh2, gmask2=mh1, save1=d2, save2=d1, backward=1, ln_y=y,
                                  ln_mean=mean, ln_rstd=rstd, ln_gamma=gm, ln_rowscale=emask, dy_out=dy, ln_dgamma=dg, ln_dbeta=db,
                                  dzb=dzb), (dz, d2, d1, dy)),
    }
    ref = {}
    print(f"B={B} N={N}: {P} pair rows; ms per launch (median of 7 after 2 warm-ups)")
    for tag, flags in VARIANTS.items():
        path = hip.LIB_PATH if not flags else lib_path(tag)
        if not os.path.exists(path):
            print(f"{tag}: library missing (run --build in the build container)")
            continue
        L = ctypes.CDLL(path)
        L.fd_edge_mlp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        line = [f"{tag:40s}"]
        for name, (d, outs) in cases.items():
            ts = []
            for i in range(9):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = L.fd_edge_mlp(ctypes.byref(d), None)
                e1.record()
                torch.cuda.synchronize()
                assert rc == 0, (tag, name, rc)
                if i >= 2:
                    ts.append(e0.elapsed_time(e1))
            ts.sort()
            same = ""
            if flags is None:
                ref[name] = [t.clone() for t in outs]
            else:
                same = " =" if all(torch.equal(a, b) for a, b in zip(ref[name], outs)) else " DIFFERS"
            line.append(f"{name} {ts[len(ts) // 2]:.3f}{same}")
        print(" | ".join(line), flush=True)


if __name__ == "__main__":
    main()
