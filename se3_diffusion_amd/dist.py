"""Data-parallel gradient exchange for the FrameDiff training step.

The path shards by whole backbones: every rank holds a full 69.8 MB fp32 replica and the
ONLY exchange step is one gradient all-reduce per step.  Instead of DDP's many ~25 MB
buckets (reference train_se3_diffusion.py:273-277, find_unused_parameters=True) all
parameter gradients live in ONE flat fp32 buffer (param.grad are views into it), so the
step issues a single RCCL all-reduce sized for xGMI (one 70 MB message; per-link bound
0.1-0.8 ms << the >= 25 ms step) and never touches unused-parameter bookkeeping: the two
parameters that never receive gradient (linear_rbf, torsion_pred.linear_3) simply
contribute zeros.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("FD_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # "nccl" is RCCL on ROCm.  FD_DIST_BACKEND=gloo: rehearsals of the multi-rank path on ONE GPU (RCCL refuses two
            # ranks on a device) -- tests/test_dist.py, `FD_DIST_BACKEND=gloo FD_FORCE_DEVICE=0 torchrun ... bench.py --gpus 2`
            backend = os.environ.get("FD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if os.environ.get("FD_FORCE_DEVICE") is not None:
            local = int(os.environ["FD_FORCE_DEVICE"])
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class FlatGrads:
    """Flat gradient buffer + one all-reduce (mean) per step."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self._dirty = False      # the buffer holds a gradient all_reduce_mean() already shipped (see rebind)

    def rebind(self):
        """p.grad must stay a view of the flat buffer: model.zero_grad() (set_to_none=True by default) or
        module.to() replace it, after which autograd would fill fresh tensors and the all-reduce would ship zeros.
        A stray gradient goes into its view, then the view is bound again: it REPLACES the view's content while the
        buffer still holds the previous step's reduced gradient (nobody called zero() on this object since: the
        model.zero_grad()-only loop; a parameter left at .grad = None gets a zeroed view), and is added otherwise
        (same rule as optim.FlatAdam.rebind)."""
        base = self.flat.data_ptr()
        for p, o in zip(self.params, self.offsets):
            if p.grad is None:
                view = self.flat[o:o + p.numel()].view_as(p)
                if self._dirty:
                    view.zero_()
                p.grad = view
            elif p.grad.data_ptr() != base + 4 * o:
                view = self.flat[o:o + p.numel()].view_as(p)
                if self._dirty:
                    view.copy_(p.grad.to(view.dtype))
                else:
                    view.add_(p.grad.to(view.dtype))
                p.grad = view

    def zero(self):
        self.flat.zero_()
        base, end = self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.flat.numel()
        for p in self.params:
            if p.grad is not None and not (base <= p.grad.data_ptr() < end):
                p.grad = None
        self._dirty = False
        self.rebind()

    def all_reduce_mean(self):
        self.rebind()
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())
        self._dirty = True


class OverlapAllReduce:
    """The gradient all-reduce in parameter groups, each started as soon as the backward pass has issued the group's last
    gradient launch (trunk.backward's on_done): heads, trunk blocks nb-1 .. 0, embedder.  The groups are contiguous
    ranges of the flat gradient buffer (named_parameters() order = embedding_layer, trunk block by block, torsion head),
    so every call is one RCCL all-reduce on a slice -- issued from the gradient side stream, i.e. behind the weight-
    gradient GEMMs that run there and, through the stream wait, behind the main stream's launches so far.

        hook = OverlapAllReduce(model, opt)        # opt: optim.FlatAdam or dist.FlatGrads over model.parameters()
        model._fd_grad_ready = hook.ready          # ScoreNetwork(accumulate_into_grad=True) forwards it to the backward
        loss.backward(); hook.finish(); opt.step()
    """

    def __init__(self, model, flat):
        self.flat = flat
        self.buf = flat.flat_g if hasattr(flat, "flat_g") else flat.flat
        offs = dict(zip((id(p) for p in flat.params), flat.offsets))
        spans = {}
        for name, p in model.named_parameters():
            if id(p) not in offs:
                continue
            if name.startswith("embedding_layer."):
                tag = "embed"
            elif name.startswith("score_model.trunk."):
                tag = int(name.split(".")[2].rsplit("_", 1)[1])      # ..._{b}
            else:
                tag = "heads"
            lo, hi = offs[id(p)], offs[id(p)] + p.numel()
            a, b = spans.get(tag, (lo, hi))
            spans[tag] = (min(a, lo), max(b, hi))
        # contiguity: the groups tile the buffer without interleaving
        order = sorted(spans.items(), key=lambda kv: kv[1][0])
        for (_, (_, hi)), (_, (lo, _)) in zip(order[:-1], order[1:]):
            assert hi <= lo, "parameter groups interleave in the flat buffer"
        self.spans = {k: (order[i][1][0], order[i + 1][1][0] if i + 1 < len(order) else self.buf.numel())
                      for i, (k, _) in enumerate(order)}
        self.handles, self.done = [], set()

    def ready(self, tag):
        if not (dist.is_initialized() and dist.get_world_size() > 1) or tag not in self.spans:
            return
        if tag in self.done:
            # a second backward before finish() (gradient accumulation) would all-reduce a chunk that already holds the
            # cross-rank sum: refuse instead of shipping a wrong gradient
            raise RuntimeError(f"OverlapAllReduce.ready({tag!r}) called twice before finish(): gradient accumulation over "
                               "several backward passes is not supported by the overlapped all-reduce (use "
                               "FlatAdam.all_reduce_mean() after the last backward instead)")
        if not self.done and hasattr(self.flat, "rebind"):
            self.flat.rebind()        # stray .grad tensors go into their views BEFORE anything is reduced
        from . import ops
        lo, hi = self.spans[tag]
        chunk = self.buf[lo:hi]
        st = ops.grad_stream(chunk.device) if chunk.is_cuda else None
        if st is not None:
            with torch.cuda.stream(st):
                self.handles.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True))
        else:
            dist.all_reduce(chunk, op=dist.ReduceOp.SUM)
        self.done.add(tag)

    def finish(self):
        """After backward(): reduce whatever the hook was not called for, wait, average."""
        if not self.done and hasattr(self.flat, "rebind"):
            self.flat.rebind()        # (hook never fired: nothing has been reduced yet, strays may still be folded in)
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return
        for tag in self.spans:
            if tag not in self.done:
                lo, hi = self.spans[tag]
                dist.all_reduce(self.buf[lo:hi], op=dist.ReduceOp.SUM)
        for h in self.handles:
            h.wait()
        self.handles, self.done = [], set()
        self.buf.div_(dist.get_world_size())
        if hasattr(self.flat, "_dirty"):
            self.flat._dirty = True


def broadcast_params(module, src=0):
    """Replicate rank-`src` parameters and buffers (what DDP does at wrap time)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


def mixed_length_schedule(steps, seed=2024, min_len=100, max_len=512, batch_size=32, max_squared_res=500000):
    """BASELINE configs[3] (cluster_time_batch on mixed lengths under DDP): the reference's DistributedTrainSampler hands
    every rank the SAME protein in a step (pdb_data_loader.py:467,483), so all ranks share N and nobody waits for a
    straggler; B = min(batch_size, max_squared_res // N^2) (data/utils.py:395, config/base.yaml:83-84).  The lengths come
    from one seeded stream that every rank reproduces: [(N, B)] * steps."""
    import numpy as np
    lens = np.random.RandomState(seed).randint(min_len, max_len + 1, size=steps)
    return [(int(n), max(1, min(batch_size, max_squared_res // (int(n) * int(n))))) for n in lens]


def shard_indices(n_items, rank, world):
    """Backbone i goes to rank i % world (reference DistributedTrainSampler: indices[rank::world])."""
    return list(range(rank, n_items, world))
