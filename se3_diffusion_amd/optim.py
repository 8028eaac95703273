"""Adam over flat buffers (fd_adam_step): the optimiser step of the training loop in one launch.

The reference trains with ``torch.optim.Adam(model.parameters(), lr=1e-4)`` (experiments/train_se3_diffusion.py:139);
over 282 parameter tensors that is 8 multi-tensor launches (0.76 ms of a 37 ms step on MI355X).  ``FlatAdam`` keeps
parameters, gradients and both moments in four flat fp32 buffers -- every ``p.data`` / ``p.grad`` is a view at a
256-byte offset (16-byte inside an ``adjacent`` group), so the GEMM kernels keep their 16-byte operand paths and the data-parallel all-reduce is one message -- and applies
the same update rule with one kernel.  ``state_dict`` / ``load_state_dict`` use torch.optim.Adam's layout, so
checkpoints written by either optimiser load into the other.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import hip

_ALIGN = 64   # elements: 256-byte aligned views


class FlatAdam:
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, adjacent=None):
        """adjacent: groups (tuples) of parameters to lay out back to back without padding, in the given order, at the
        position of the group's first member (ScoreNetwork.flat_layout_groups(): weights that one kernel reads as one
        matrix, e.g. IPA's [linear_b ; down_z]).  The state_dict order stays that of `params`."""
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        assert all(p.dtype == torch.float32 for p in self.params), "fp32 parameters only"
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        dev = self.params[0].device
        group_of = {}
        for grp in (adjacent or ()):
            grp = [p for p in grp if p.requires_grad]
            assert all(p.numel() % 4 == 0 for p in grp[:-1]), "grouped parameters must keep their successors 16-byte aligned"
            for p in grp:
                group_of[id(p)] = grp
        placed, off = {}, 0
        for p in self.params:
            if id(p) in placed:
                continue
            for q in group_of.get(id(p), [p]):       # (a group is placed when its first member comes up)
                if id(q) in placed:
                    continue
                placed[id(q)] = off
                off += q.numel()
            off = (off + _ALIGN - 1) // _ALIGN * _ALIGN
        self.offsets = [placed[id(p)] for p in self.params]
        self.numel = off
        self.flat_p = torch.zeros(off, device=dev)
        self.flat_g = torch.zeros(off, device=dev)
        self.exp_avg = torch.zeros(off, device=dev)
        self.exp_avg_sq = torch.zeros(off, device=dev)
        for p, o in zip(self.params, self.offsets):
            view = self.flat_p[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[o:o + p.numel()].view_as(p)
        self.t = 0
        self._dirty = False      # flat_g holds a gradient that step() / all_reduce_mean() already consumed (see rebind)

    def rebind(self):
        """Make every p.data / p.grad a view of the flat buffers again.  Anything that replaces them behind the
        optimiser's back -- model.zero_grad() (set_to_none=True by default), module.to() / _apply -- would otherwise
        leave flat_g all-zero while autograd fills fresh tensors: the step would run on zeros without an error.
        A stray parameter value is copied into its view.  A stray gradient follows torch's semantics for a parameter
        whose .grad was None: the fresh tensor IS the gradient.  While the flat buffer still holds the gradient the last
        step() / all_reduce_mean() consumed (`_dirty`: nobody called zero_grad() on THIS object since -- the
        model.zero_grad()-only training loop), the stray gradient therefore REPLACES its view and a parameter left at
        .grad = None gets a zeroed view; after a zero_grad() it is added (nothing stale to overwrite, and a caller may
        have accumulated into the view before detaching it)."""
        bp, bg = self.flat_p.data_ptr(), self.flat_g.data_ptr()
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if p.data_ptr() == bp + 4 * o and g is not None and g.data_ptr() == bg + 4 * o:
                continue                                   # (the common case: two pointer reads per parameter)
            n = p.numel()
            if p.data.data_ptr() != self.flat_p.data_ptr() + 4 * o:
                view = self.flat_p[o:o + n].view_as(p)
                view.copy_(p.data)
                p.data = view
            if p.grad is None:
                view = self.flat_g[o:o + n].view_as(p)
                if self._dirty:
                    view.zero_()
                p.grad = view
            elif p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                view = self.flat_g[o:o + n].view_as(p)
                if self._dirty:
                    view.copy_(p.grad.to(view.dtype))
                else:
                    view.add_(p.grad.to(view.dtype))
                p.grad = view

    # -- gradient buffer (dist.FlatGrads interface) ------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()
        for p in self.params:          # a stray gradient (see rebind) is dropped with the rest
            if p.grad is not None and not (self.flat_g.data_ptr() <= p.grad.data_ptr() < self.flat_g.data_ptr() + 4 * self.numel):
                p.grad = None
        self._dirty = False
        self.rebind()

    zero = zero_grad

    def all_reduce_mean(self):
        self.rebind()
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM)
            self.flat_g.div_(dist.get_world_size())
        self._dirty = True

    # -- update ------------------------------------------------------------------------------------------------------
    def step(self):
        self.rebind()
        self.t += 1
        b1, b2 = self.betas
        hip.get_lib().call("fd_adam_step", self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.numel,
                           self.lr, b1, b2, self.eps, 1.0 - b1 ** self.t, 1.0 - b2 ** self.t)
        self._dirty = True

    # -- torch.optim.Adam-compatible checkpoints (data/utils.py:353-362 saves optimizer.state_dict()) ----------------
    def state_dict(self):
        state = {}
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.t)),
                        "exp_avg": self.exp_avg[o:o + n].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + n].view_as(p).clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        self.lr, self.betas, self.eps = float(group["lr"]), tuple(float(b) for b in group["betas"]), float(group["eps"])
        assert not group.get("weight_decay", 0) and not group.get("amsgrad", False), "plain Adam only"
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            st = sd["state"].get(i)
            if st is None:
                continue
            n = p.numel()
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            self.t = int(float(st["step"]))
