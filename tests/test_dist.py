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


def test_mixed_length_schedule_is_rank_invariant():
    """configs[3]: every rank derives the same (N, B) per step, B follows the reference's length_batching rule"""
    from se3_diffusion_amd import dist as fdist
    a = fdist.mixed_length_schedule(50)
    assert a == fdist.mixed_length_schedule(50) and len(a) == 50
    for n, b in a:
        assert 100 <= n <= 512 and b == max(1, min(32, 500000 // (n * n)))
    assert len({n for n, _ in a}) > 20 and min(b for _, b in a) == 1 and max(b for _, b in a) == 32


def test_shard_indices():
    from se3_diffusion_amd import dist as fdist
    got = sorted(i for r in range(4) for i in fdist.shard_indices(10, r, 4))
    assert got == list(range(10))


# ---------------------------------------------------------------------------------------------------------------------
# the real ScoreNetwork (kernel sources under the host interpreter, 1 block, N = 8): 2 ranks x B = 1 == 1 rank x B = 2
# ---------------------------------------------------------------------------------------------------------------------
def _sn_setup(emu_path):
    from oracle import framediff_oracle as fo
    from se3_diffusion_amd import hip, train_step as ts
    from se3_diffusion_amd.model.score_network import ScoreNetwork
    hip._TEST_OVERRIDE = hip.FdLib(emu_path)
    conf = dict(fo.CONF, num_blocks=1)
    model = ScoreNetwork(ts.base_model_conf(1), diffuser=None)
    model.load_state_dict(fo.synth_params(seed=3, conf=conf), strict=True)
    model.train()
    batch = ts.synthetic_batch(2, 8, "cpu", seed=9)
    gt37, _ = fo.backbone_atoms(batch["rigids_0"][..., :4], batch["rigids_0"][..., 4:], batch["torsion_angles_sin_cos"][..., 2, :])
    return model, batch, gt37, ts


def _sn_loss(ts, model, batch, gt37, sl):
    b = {k: v[sl] for k, v in batch.items()}
    return ts.dsm_loss(b, model(b), gt37[sl])


def _sn_worker(rank, world, port, q, emu_path, mode):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from se3_diffusion_amd import dist as fdist
    fdist.init_from_env(backend="gloo")
    model, batch, gt37, ts = _sn_setup(emu_path)
    sl = slice(rank, rank + 1)                                  # backbone i -> rank i (shard_indices(2, rank, 2))
    assert fdist.shard_indices(2, rank, world) == [rank]
    names = [n for n, _ in model.named_parameters()]
    if mode == "flat":
        from se3_diffusion_amd.optim import FlatAdam
        # (linear_b / down_z of every IPA block back to back: the kernels read and accumulate them as one [40, 128] matrix)
        opt = FlatAdam(model.parameters(), lr=1e-3, adjacent=model.flat_layout_groups())
        pb, pz = model.score_model.trunk["ipa_0"].linear_b.weight, model.score_model.trunk["ipa_0"].down_z.weight
        assert pb.data_ptr() + 4 * pb.numel() == pz.data_ptr() and pb.grad.data_ptr() + 4 * pb.numel() == pz.grad.data_ptr()
        model.accumulate_into_grad = True                       # gradients are written straight into the all-reduce buffer
        hook = fdist.OverlapAllReduce(model, opt)               # per-group all-reduce from inside the backward (bench.py, N > 1)
        model._fd_grad_ready = hook.ready
        assert set(hook.spans) == {"embed", 0, "heads"} and sum(b - a for a, b in hook.spans.values()) == opt.numel
        opt.zero_grad()
        _sn_loss(ts, model, batch, gt37, sl).backward()
        assert hook.done == {"embed", 0, "heads"}
        hook.finish()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        opt.step()
    else:
        # the reference's own wrapping (train_se3_diffusion.py:273-277): DistributedDataParallel with
        # find_unused_parameters=True (linear_rbf / torsion_pred.linear_3 never receive gradient)
        ddp = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        opt.zero_grad()
        _sn_loss(ts, ddp, batch, gt37, sl).backward()
        grads = {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()}
        opt.step()
    q.put((rank, {n: g.double().numpy() for n, g in grads.items()},
           torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _sn_single(emu_path):
    from se3_diffusion_amd import hip
    model, batch, gt37, ts = _sn_setup(emu_path)
    # DP averages the per-rank losses (each normalised by its own example count: B = 1 per rank)
    loss = 0.5 * (_sn_loss(ts, model, batch, gt37, slice(0, 1)) + _sn_loss(ts, model, batch, gt37, slice(1, 2)))
    loss.backward()
    hip._TEST_OVERRIDE = None
    return {n: (p.grad.double() if p.grad is not None else torch.zeros_like(p).double()) for n, p in model.named_parameters()}


@pytest.mark.parametrize("mode", ["flat", "ddp"])
def test_score_network_data_parallel_matches_single_process(emu_lib, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sn_worker, args=(r, 2, port, q, emu_lib.path, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (g, torch.tensor(w)) for r, g, w in (q.get(timeout=600) for _ in range(2))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(res[0][1], res[1][1])                    # replicas stay bit-identical after the optimiser step
    ref = _sn_single(emu_lib.path)
    # the all-reduced gradient == the gradient of the single-process two-backbone loss, to fp32 round-off (compared
    # before Adam: its normalised update turns the round-off of analytically-zero gradients into +-lr steps)
    bad = []
    for n, g in ref.items():
        got = torch.tensor(res[0][0][n])
        err = float((got - g).abs().max())
        if err > 2e-5 * float(g.abs().max()) + 1e-7:
            bad.append((n, err, float(g.abs().max())))
    assert not bad, bad[:8]


# ---------------------------------------------------------------------------------------------------------------------
# GPU tier: the same check with the PRODUCT library on CUDA tensors -- two processes share device 0 (RCCL refuses two ranks
# on one GPU, so the collective backend is gloo, which stages CUDA tensors through pinned host memory).  This is the only
# place short of an 8-GPU node where OverlapAllReduce.ready() takes its device branch: async all-reduce of a flat-buffer
# slice issued from the gradient side stream while the backward pass is still enqueueing kernels (dist.py).
# ---------------------------------------------------------------------------------------------------------------------
_GPU_B, _GPU_N, _GPU_BLOCKS = 2, 24, 2


def _sn_setup_gpu():
    from oracle import framediff_oracle as fo
    from se3_diffusion_amd import train_step as ts
    from se3_diffusion_amd.model.score_network import ScoreNetwork
    conf = dict(fo.CONF, num_blocks=_GPU_BLOCKS)
    model = ScoreNetwork(ts.base_model_conf(_GPU_BLOCKS), diffuser=None)
    model.load_state_dict(fo.synth_params(seed=3, conf=conf), strict=True)
    model = model.cuda().train()
    batch = ts.synthetic_batch(_GPU_B, _GPU_N, "cuda", seed=9)
    gt37, _ = ts.backbone_atoms(batch["rigids_0"], batch["torsion_angles_sin_cos"][..., 2, :])
    return model, batch, gt37


def _sn_loss_gpu(model, batch, gt37, sl):
    from se3_diffusion_amd import loss as floss
    b = {k: v[sl].contiguous() for k, v in batch.items()}
    return floss.dsm_loss(b, model(b), gt37[sl].contiguous())


def _sn_worker_gpu(rank, world, port, q, overlap):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    from se3_diffusion_amd import dist as fdist, ops
    from se3_diffusion_amd.optim import FlatAdam
    fdist.init_from_env(backend="gloo")
    try:
        probe = torch.ones(4, device="cuda") * (rank + 1)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        assert float(probe[0]) == 3.0
    except RuntimeError as e:                                   # a torch build whose gloo cannot take device tensors
        q.put((rank, "unsupported: " + str(e)[:200], None))
        return
    model, batch, gt37 = _sn_setup_gpu()
    opt = FlatAdam(model.parameters(), lr=1e-3, adjacent=model.flat_layout_groups())
    model.accumulate_into_grad = True
    hook = None
    if overlap:
        hook = fdist.OverlapAllReduce(model, opt)
        model._fd_grad_ready = hook.ready
        assert ops.grad_stream(torch.device("cuda", 0)) is not None          # the device branch of ready() is the one taken
    sl = slice(rank, rank + 1)
    for it in range(2):                                         # twice: the second pass reuses the bound views / streams
        opt.zero_grad()
        _sn_loss_gpu(model, batch, gt37, sl).backward()
        if hook is not None:
            assert hook.done == {"embed", "heads"} | set(range(_GPU_BLOCKS)) and len(hook.handles) == len(hook.done)
            hook.finish()
        else:
            opt.all_reduce_mean()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().double().cpu().numpy() for n, p in model.named_parameters()}
        if it == 0 and hook is not None:
            # a second backward before finish() must be refused, not silently re-reduced
            hook.done.add("heads")
            try:
                hook.ready("heads")
                raise AssertionError("ready() accepted a second call for one tag")
            except RuntimeError:
                pass
            hook.done.clear()
    opt.step()
    torch.cuda.synchronize()
    q.put((rank, grads, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double().cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [True, False])
def test_score_network_data_parallel_gpu(hip_lib, overlap):
    """2 ranks x 1 backbone on device 0 == 1 process x 2 backbones: gradients after the (overlapped) all-reduce, replicas
    bit-identical after FlatAdam.step()."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sn_worker_gpu, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
    if any(isinstance(g, str) for _, g, _ in got):
        pytest.skip(f"gloo in this torch build does not take device tensors: {got[0][1]}")
    assert all(p.exitcode == 0 for p in procs)
    res = {r: (g, torch.tensor(w)) for r, g, w in got}
    assert torch.equal(res[0][1], res[1][1])                    # replicas stay bit-identical after the optimiser step
    model, batch, gt37 = _sn_setup_gpu()
    loss = 0.5 * (_sn_loss_gpu(model, batch, gt37, slice(0, 1)) + _sn_loss_gpu(model, batch, gt37, slice(1, 2)))
    loss.backward()
    bad = []
    for n, p in model.named_parameters():
        g = (p.grad if p.grad is not None else torch.zeros_like(p)).double().cpu()
        err = float((torch.tensor(res[0][0][n]) - g).abs().max())
        if err > 1e-4 * float(g.abs().max()) + 1e-6:             # fp32 round-off of two summation orders (split-K atomics)
            bad.append((n, err, float(g.abs().max())))
    assert not bad, bad[:8]


_RCCL_ONE_RANK = r'''
import json, os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)          # "nccl" IS RCCL on ROCm
from se3_diffusion_amd import ops, train_step as ts
from se3_diffusion_amd.model.score_network import ScoreNetwork
from se3_diffusion_amd.optim import FlatAdam
model = ScoreNetwork(ts.base_model_conf(1), diffuser=None).cuda()
opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=model.flat_layout_groups())
g = torch.Generator(device="cuda").manual_seed(0)
opt.flat_g.normal_(generator=g)
ref = opt.flat_g.clone()
# the step's collective: ONE all-reduce over the flat fp32 gradient buffer (the full-depth model's is 69.8 MB), in place
full = torch.randn(17_446_190, device="cuda", generator=g)
full_ref = full.clone()
dist.all_reduce(full, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    dist.all_reduce(full, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
# the overlapped form's device branch: async all-reduce of a SLICE issued from the gradient side stream
st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    h = dist.all_reduce(opt.flat_g[1000:500000], op=dist.ReduceOp.SUM, async_op=True)
h.wait()
torch.cuda.current_stream().wait_stream(st)
for t in list(model.parameters())[:8]:
    dist.broadcast(t.data, src=0)
dist.barrier()
torch.cuda.synchronize()
ok = bool(torch.equal(opt.flat_g, ref) and torch.equal(full, full_ref))
print(json.dumps({"ok": ok, "backend": dist.get_backend(), "allreduce_70MB_ms": round(ms, 3),
                  "nccl_version": ".".join(str(v) for v in torch.cuda.nccl.version())}))
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_rccl_one_rank_gpu(hip_lib):
    """RCCL itself (not gloo) under the step's collectives, as far as one GPU goes: a 1-rank `nccl` process group runs the flat
    69.8 MB gradient all-reduce in place, an asynchronous all-reduce of a slice issued from a side stream (the overlapped
    form's device branch), the parameter broadcast and a barrier; values must come back unchanged.  (RCCL refuses two ranks on
    one device, so the 2-rank tests above use gloo; the N-GPU run is the driver's.)"""
    import json
    import subprocess
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, ROOT, str(_free_port())], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print("[rccl one rank]", d)
    assert d["ok"] and d["backend"] == "nccl"
