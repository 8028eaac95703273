"""Build libfd_hip.so (gfx950) from se3_diffusion_amd/csrc/*.hip with hipcc.

In-tree, incremental (per-source objects under csrc/_obj, rebuilt when the
source or any header is newer), parallel.  `python -m se3_diffusion_amd.build`.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "lib", "libfd_hip.so")
ROOT = os.path.dirname(HERE)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-I", os.path.join(CSRC, "gfx950"), "-I", CSRC, "-I", os.path.join(ROOT, "include"),
    "-Wno-unused-result",
]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _hip_includes(src):
    """newest mtime of the .hip files a source includes (fd_edge_mlp_w8.hip instantiates fd_edge_mlp.hip's kernel)"""
    import re
    m = 0.0
    with open(src) as f:
        for inc in re.findall(r'#include "([^"]+\.hip)"', f.read()):
            q = os.path.join(os.path.dirname(src), inc)
            if os.path.exists(q):
                m = max(m, os.path.getmtime(q))
    return m


def _headers_mtime():
    m = 0.0
    for d in (CSRC, os.path.join(CSRC, "gfx950"), os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def build(verbose=True, force=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hm = _headers_mtime()
    jobs = []
    objs = []
    for src in _sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hm, _hip_includes(src)):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for src, r in ex.map(cc, jobs):
                if verbose and (r.stderr.strip() or r.returncode):
                    sys.stderr.write(r.stderr)
                if r.returncode:
                    raise RuntimeError(f"hipcc failed on {src}")
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr)
            raise RuntimeError("link failed")
    if verbose:
        print(f"[fd build] {len(jobs)} compiled -> {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
