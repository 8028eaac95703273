"""Where the spilled VGPRs of the register-chained kernels live: compiles a kernel source to gfx950 ISA and reports, per kernel,
the register / scratch figures of -Rpass-analysis=kernel-resource-usage and the position of every scratch_* instruction relative
to the kernel's MFMA stream (before the first MFMA = once per launch, between MFMAs = inside the tile loop, after the last).

    python tools/spill_report.py [csrc file ...]  > profiles/rNN_edge_mlp_spills.txt      (build container; no GPU needed)
"""
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from se3_diffusion_amd import build  # noqa: E402


def report(src):
    asm = "/tmp/fd_spill_report.s"
    r = subprocess.run([build.HIPCC, *build.FLAGS, "-Rpass-analysis=kernel-resource-usage", "-S", "--cuda-device-only", "-o", asm, src],
                       capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    res, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
        for key in ("VGPRs:", "AGPRs:", "VGPRs Spill:", "SGPRs Spill:", "ScratchSize [bytes/lane]:", "LDS Size [bytes/block]:", "Occupancy [waves/SIMD]:"):
            if cur is not None and key in line:
                cur[key] = line.split(key)[1].split()[0]
    text = open(asm).read()
    print(f"## {os.path.relpath(src, ROOT)}")
    for chunk in re.split(r"\n(?=_Z\w+:\s)", text):
        name = chunk.split(":")[0].strip()
        if name not in res or "v_mfma" not in chunk:
            continue
        nm, ops = 0, []
        for line in chunk.split("\n"):
            if "v_mfma" in line:
                nm += 1
            if "scratch_" in line:
                ops.append((nm, line.strip().split()[0].replace("scratch_", "")))
            if "s_endpgm" in line:
                break
        def c(sel):
            return dict(Counter(o for n_, o in ops if sel(n_))) or "-"
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        dem = dem.replace("(anonymous namespace)::", "").replace("(FdEdgeMlpDesc)", "")
        k = res[name]
        print(f"{dem}: {k.get('VGPRs:')} VGPRs + {k.get('AGPRs:')} AGPRs, {k.get('VGPRs Spill:')} VGPRs / {k.get('SGPRs Spill:')} SGPRs spilled, "
              f"scratch {k.get('ScratchSize [bytes/lane]:')} B/lane, LDS {k.get('LDS Size [bytes/block]:')} B, {nm} MFMAs per tile")
        print(f"    scratch ops before the first MFMA (once per launch): {c(lambda n_: n_ == 0)}")
        print(f"    between MFMAs (per tile):                            {c(lambda n_: 0 < n_ < nm)}")
        print(f"    after the last MFMA (per tile):                      {c(lambda n_: n_ >= nm)}")
    print()


if __name__ == "__main__":
    files = sys.argv[1:] or [os.path.join(build.CSRC, f) for f in ("fd_edge_mlp.hip", "fd_edge_mlp_w8.hip")]
    print("# spilled registers of the fused edge-transition kernels and where their scratch traffic sits (tools/spill_report.py;\n"
          "# template arguments: BWD, ZB, TRAIN, LNB, ZBW).  `store_*` before the first MFMA = loop-invariant addresses parked once per launch;\n"
          "# inside the tile loop only 8-byte reloads of those remain.\n")
    for f in files:
        report(os.path.abspath(f))
