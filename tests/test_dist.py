"""Multi-GPU path on CPU: world_size-2 gloo processes exercise se3_diffusion_amd.dist exactly as bench.py
uses it (flat gradient buffer, one all-reduce, parameter broadcast, backbone sharding).  The data path has
no other collective (whole backbones are independent units)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, emu_path=None):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from se3_diffusion_amd import dist as fdist
    r, w, _ = fdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                     # ranks start different ...
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    fdist.broadcast_params(model)                     # ... and are made identical
    if emu_path is None:
        flat = fdist.FlatGrads(model.parameters())
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    else:
        # bench.py's configuration: FlatAdam owns parameters, gradients and moments as flat buffers (its kernel runs
        # under the host interpreter here)
        from se3_diffusion_amd import hip
        from se3_diffusion_amd.optim import FlatAdam
        hip._TEST_OVERRIDE = hip.FdLib(emu_path)
        opt = flat = FlatAdam(model.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 6, generator=g)
    Y = torch.randn(8, 3, generator=g)
    idx = fdist.shard_indices(8, rank, world)         # backbone i -> rank i % world
    for _ in range(3):
        flat.zero()
        loss = ((model(X[idx]) - Y[idx]) ** 2).mean()
        loss.backward()
        base = (flat.flat if emu_path is None else flat.flat_g).data_ptr()
        assert all(p.grad.data_ptr() >= base for p in model.parameters())  # grads live in the flat buffer
        flat.all_reduce_mean()
        opt.step()
    q.put((rank, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double().tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _single():
    torch.manual_seed(100)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 6, generator=g)
    Y = torch.randn(8, 3, generator=g)
    for _ in range(3):
        opt.zero_grad()
        # mean over the two half-batches == DP average of per-rank means
        loss = 0.5 * (((model(X[0::2]) - Y[0::2]) ** 2).mean() + ((model(X[1::2]) - Y[1::2]) ** 2).mean())
        loss.backward()
        opt.step()
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()])


def test_flat_adam_data_parallel_matches_single_process(emu_lib):
    _dp_check(emu_lib.path)


def test_flat_grad_allreduce_matches_single_process():
    _dp_check(None)


def _dp_check(emu_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, emu_path)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: torch.tensor(v, dtype=torch.float64).float() for r, v in (q.get(timeout=240) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.allclose(res[0], res[1], atol=0, rtol=0)          # replicas stay bit-identical
    assert torch.allclose(res[0], _single(), atol=1e-6)            # == single-process large batch


def test_shard_indices():
    from se3_diffusion_amd import dist as fdist
    got = sorted(i for r in range(4) for i in fdist.shard_indices(10, r, 4))
    assert got == list(range(10))
