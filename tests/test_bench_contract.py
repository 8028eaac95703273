"""bench.py's output contract (one JSON line with the driver's fields + `roofline` + `cpu_baseline`) on a small configuration,
and a rehearsal of its multi-rank path: `python bench.py --gpus 2` starting its own two ranks on ONE GPU (FD_DIST_BACKEND=gloo,
FD_FORCE_DEVICE=0 -- RCCL refuses two ranks on a device), i.e. the per-rank data, parameter broadcast, flat all-reduce,
barrier-bracketed timing and max-over-ranks code that `--gpus N` runs on an 8-GPU node."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
SMALL = ["--n-res", "32", "--batch", "2", "--blocks", "1", "--steps", "2", "--warmup", "1", "--no-sampling"]


def _line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _check(d, n):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == n and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "residues/s" and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - n * 2 * 32 / d["ms_per_step"] * 1e3) < 0.02 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "workload" in d["config"] and "model" not in d["config"]
    sp = d["config"]["step_ms_spread"]
    assert sp["min"] <= sp["median"] <= sp["max"] and 0 <= sp["slowest_step"] < d["steps"] and sp["device_allocations_in_timed_region"] >= 0


def test_bench_line_single(hip_lib):
    env = dict(os.environ, FD_BENCH_PRIME="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, "--cpu-sample-batch", "1"], capture_output=True,
                       text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    _check(d, 1)
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb


def test_bench_two_ranks_on_one_gpu(hip_lib):
    """plain `python bench.py --gpus 2`: bench.py starts its two ranks itself (spawn_ranks -> torch.distributed.run); gloo and a
    forced device because RCCL refuses two ranks on one GPU -- on an 8-GPU node the same command runs over RCCL."""
    env = dict(os.environ, FD_BENCH_PRIME="1", FD_DIST_BACKEND="gloo", FD_FORCE_DEVICE="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _line(r.stdout)
    _check(d, 2)
    c = d["config"]
    assert c["ranks"] == 2 and c["collective_backend"] == "gloo" and c["rccl_ranks"] == 0 and len(c["ms_per_step_by_rank"]) == 2


def test_bench_under_torchrun(hip_lib):
    """the driver's own launch form for N > 1 (torch.distributed.run around bench.py --gpus N) still works and is not re-spawned"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, FD_BENCH_PRIME="1", FD_DIST_BACKEND="gloo", FD_FORCE_DEVICE="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    _check(_line(r.stdout), 2)


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_bench_two_ranks_over_rccl(hip_lib, overlap):
    """(overlap = "1": FD_DP_OVERLAP=1 -- the slice all-reduces of dist.OverlapAllReduce issued from the gradient side stream while
    the backward is still enqueueing -- so that the first time that path meets RCCL is not in front of the driver.)
    `python bench.py --gpus 2` over nccl (= RCCL over xGMI) on a box that HAS two GPUs: the first multi-rank RCCL init, the
    parameter broadcast, the flat-gradient all-reduce and the barrier-bracketed timing with one rank per device -- the path
    the driver's SCALE run takes (reference DDP launch: experiments/train_se3_diffusion.py:83-91,273-277).  Skipped on the
    one-GPU boxes this repository is developed on (RCCL has so far only run with ONE rank: tests/test_dist.py)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the gpurun boxes have one)")
    env = dict(os.environ, FD_BENCH_PRIME="1", HSA_ENABLE_IPC_MODE_LEGACY="0", FD_DP_OVERLAP=overlap)
    for k in ("WORLD_SIZE", "FD_DIST_BACKEND", "FD_FORCE_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _line(r.stdout)
    _check(d, 2)
    c = d["config"]
    assert c["ranks"] == 2 and c["collective_backend"] == "nccl" and c["rccl_ranks"] == 2 and len(c["ms_per_step_by_rank"]) == 2
    assert bool(c.get("dp_overlap", False)) == (overlap == "1")


def test_bench_takes_world_size_from_the_launcher(hip_lib):
    """torchrun --nproc-per-node=2 bench.py WITHOUT --gpus 2: the launcher's WORLD_SIZE wins (a warning, not an AssertionError
    on every rank)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, FD_BENCH_PRIME="1", FD_DIST_BACKEND="gloo", FD_FORCE_DEVICE="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), *SMALL],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "measuring 2 ranks" in r.stderr
    _check(_line(r.stdout), 2)
