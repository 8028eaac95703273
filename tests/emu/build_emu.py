"""TEST INFRASTRUCTURE ONLY: compile se3_diffusion_amd/csrc/*.hip with g++
against the fiber SIMT interpreter in this directory -> tests/emu/_build/libfd_emu.so.
The product package never loads this library."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "se3_diffusion_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libfd_emu.so")
FLAGS = ["-O1", "-g0", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-strict-aliasing",
         "-I", os.path.join(HERE, "include"), "-I", CSRC, "-I", os.path.join(ROOT, "include"),
         "-Wno-unused-result", "-Wno-attributes", "-Wno-unknown-pragmas"]


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


def build(verbose=True, force=False):
    os.makedirs(OUT, exist_ok=True)
    hm = 0.0
    for d in (CSRC, os.path.join(HERE, "include"), os.path.join(HERE, "include", "hip"), os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                hm = max(hm, os.path.getmtime(os.path.join(d, f)))
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    srcs.append(os.path.join(HERE, "hipemu.cpp"))
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s).rsplit(".", 1)[0] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hm, _hip_includes(s) if s.endswith(".hip") else 0.0):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        return s, subprocess.run(["g++", *FLAGS, "-x", "c++", "-c", s, "-o", o], capture_output=True, text=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for s, r in ex.map(cc, jobs):
                if r.returncode:
                    sys.stderr.write(r.stderr)
                    raise RuntimeError(f"g++ failed on {s}")
                elif verbose and r.stderr.strip():
                    sys.stderr.write(r.stderr[:4000])
    if jobs or not os.path.exists(LIB):
        r = subprocess.run(["g++", "-shared", "-fPIC", *objs, "-o", LIB], capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr)
            raise RuntimeError("emu link failed")
    if verbose:
        print(f"[fd emu] {len(jobs)} compiled -> {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
