"""Build tools/experiments/libfd_experiments.so (gfx950): the kernels that lost their A/B and left the product library.
    python tools/experiments/build.py
Never called by __graft_entry__.build(), never loaded by se3_diffusion_amd."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "se3_diffusion_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LIB = os.path.join(HERE, "libfd_experiments.so")


def build():
    srcs = sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hip"))
    srcs.append(os.path.join(CSRC, "fd_api.hip"))        # fd_last_error / FD_CHECK_* plumbing
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", HERE,
           "-I", os.path.join(CSRC, "gfx950"), "-I", CSRC, "-I", os.path.join(ROOT, "include"), *srcs, "-o", LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    sys.stderr.write(r.stderr)
    if r.returncode:
        raise RuntimeError("hipcc failed")
    print("built", LIB)
    return LIB


if __name__ == "__main__":
    build()
