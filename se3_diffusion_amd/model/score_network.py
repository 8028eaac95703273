"""Drop-in for the reference ``model/score_network.py``: same public names
(get_index_embedding, get_timestep_embedding, Embedder, ScoreNetwork), same
constructor / forward signatures and state_dict keys -- the forward and backward of
the whole network run as hand-written HIP kernels (se3_diffusion_amd/trunk.py).

ScoreNetwork.forward(input_feats) -> dict(psi, rot_score, trans_score, rigids, atom37, atom14)
(reference score_network.py:170-215).  Differentiable w.r.t. the parameters.
"""
import functools as fn
import math

import torch
from torch import nn

from .. import trunk
from ..network import check_conf
from . import ipa_pytorch

Tensor = torch.Tensor


def get_index_embedding(indices, embed_size, max_len=2056):
    """Sin/cos index embedding (host helper; the forward pass builds it in fd_node/edge_feats)."""
    k = torch.arange(embed_size // 2, device=indices.device)
    ang = indices[..., None] * math.pi / (max_len ** (2 * k[None] / embed_size))
    return torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)


def get_timestep_embedding(timesteps, embedding_dim, max_positions=10000):
    """Sinusoidal timestep embedding (host helper; see fd_node_feats)."""
    assert len(timesteps.shape) == 1
    half = embedding_dim // 2
    freqs = torch.exp(torch.arange(half, dtype=torch.float32, device=timesteps.device)
                      * -(math.log(max_positions) / (half - 1)))
    emb = (timesteps * max_positions).float()[:, None] * freqs[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1), mode="constant")
    assert emb.shape == (timesteps.shape[0], embedding_dim)
    return emb


class Embedder(nn.Module):
    """Parameter container of the node/edge embedder MLPs (reference score_network.py:49-95)."""

    def __init__(self, model_conf):
        super().__init__()
        self._model_conf = model_conf
        self._embed_conf = model_conf.embed
        idx = self._embed_conf.index_embed_size
        node_in = idx + 1 + idx
        edge_in = (idx + 1) * 2 + idx
        if self._embed_conf.embed_self_conditioning:
            edge_in += self._embed_conf.num_bins
        ns, es = model_conf.node_embed_size, model_conf.edge_embed_size
        self.node_embedder = nn.Sequential(nn.Linear(node_in, ns), nn.ReLU(), nn.Linear(ns, ns), nn.ReLU(),
                                           nn.Linear(ns, ns), nn.LayerNorm(ns))
        self.edge_embedder = nn.Sequential(nn.Linear(edge_in, es), nn.ReLU(), nn.Linear(es, es), nn.ReLU(),
                                           nn.Linear(es, es), nn.LayerNorm(es))
        self.timestep_embedder = fn.partial(get_timestep_embedding, embedding_dim=idx)
        self.index_embedder = fn.partial(get_index_embedding, embed_size=idx)


_FEAT_KEYS = ("res_mask", "fixed_mask", "seq_idx", "t", "sc_ca_t", "rigids_t", "torsion_angles_sin_cos")
_OUT_KEYS = ("psi", "rot_score", "trans_score", "rigids", "atom37", "atom14")


class _ScoreNetFn(torch.autograd.Function):
    """One autograd node for the whole network: forward = trunk.forward, backward = trunk.backward."""

    @staticmethod
    def forward(ctx, module, feats, names, need_grad, *params):
        P = dict(zip(names, params))
        bool_mask = (not module.training) and not need_grad       # nn.TransformerEncoder fast-path semantics
        out, sv = trunk.forward(P, feats, module._num_blocks, module._dconf, tfmr_bool_mask=bool_mask,
                                save=need_grad, cache=getattr(module, "_fd_static", None),
                                sc_ca_out=None if need_grad else getattr(module, "_fd_sc_ca_out", None))
        ctx.sv, ctx.P, ctx.names, ctx.module = sv, P, names, module
        res = tuple(out[k] for k in _OUT_KEYS)
        ctx.mark_non_differentiable(res[5])
        return res

    @staticmethod
    def backward(ctx, d_psi, d_rot, d_ts, d_rig, d_a37, d_a14):
        P, names = ctx.P, ctx.names
        d_out = dict(psi=d_psi, rot_score=d_rot, trans_score=d_ts, rigids=d_rig, atom37=d_a37)
        if ctx.module.accumulate_into_grad and all(p.grad is not None for p in P.values()):
            # every kernel of the backward ACCUMULATES (+=) its parameter gradient, so when the caller pre-allocates
            # .grad (e.g. dist.FlatGrads: one flat buffer for the single RCCL all-reduce) the gradients are written
            # in place -- no 282 zero-fills + 282 autograd accumulation kernels per step
            G = {k: v.grad for k, v in P.items()}
            trunk.backward(P, G, ctx.sv, d_out, on_done=getattr(ctx.module, "_fd_grad_ready", None))
            ctx.sv = None
            return (None, None, None, None) + tuple(None for _ in names)
        G = {k: torch.zeros_like(v) for k, v in P.items()}
        trunk.backward(P, G, ctx.sv, d_out)
        ctx.sv = None
        return (None, None, None, None) + tuple(G[n] for n in names)


class ScoreNetwork(nn.Module):

    def __init__(self, model_conf, diffuser):
        super().__init__()
        check_conf(model_conf)
        self._model_conf = model_conf
        self.embedding_layer = Embedder(model_conf)
        self.diffuser = diffuser
        self.score_model = ipa_pytorch.IpaScore(model_conf, diffuser)
        self._num_blocks = model_conf.ipa.num_blocks
        # opt-in (bench.py / dist.FlatGrads): write parameter gradients straight into pre-allocated .grad tensors.
        # Leave False under torch DDP, whose reducer hooks the autograd accumulation.
        self.accumulate_into_grad = False
        self._dconf = _diffuser_consts(model_conf, diffuser)

    def _apply_mask(self, aatype_diff, aatype_0, diff_mask):
        return diff_mask * aatype_diff + (1 - diff_mask) * aatype_0

    def flat_layout_groups(self):
        """Parameter groups that a flat optimiser (optim.FlatAdam(adjacent=...)) should lay out back to back: IPA's
        linear_b / down_z weights and biases are read by the kernels as ONE [40, 128] matrix / [40] vector."""
        groups = []
        named = dict(self.named_parameters())
        b = 0
        while f"score_model.trunk.ipa_{b}.linear_b.weight" in named:
            pre = f"score_model.trunk.ipa_{b}"
            groups.append((named[f"{pre}.linear_b.weight"], named[f"{pre}.down_z.weight"]))
            groups.append((named[f"{pre}.linear_b.bias"], named[f"{pre}.down_z.bias"]))
            proj = ("linear_q", "linear_kv", "linear_q_points", "linear_kv_points")     # one [6816, 256] projection of s
            groups.append(tuple(named[f"{pre}.{n}.weight"] for n in proj))
            groups.append(tuple(named[f"{pre}.{n}.bias"] for n in proj))
            b += 1
        return groups

    def forward(self, input_feats):
        """input_feats: res_mask[B,N], fixed_mask[B,N], seq_idx[B,N], t[B], sc_ca_t[B,N,3],
        rigids_t[B,N,7], torsion_angles_sin_cos[B,N,7,2] (extra keys ignored)."""
        feats = {k: input_feats[k] for k in _FEAT_KEYS}
        names, params = zip(*self.named_parameters())
        # (grad mode is off inside Function.forward, so the decision is taken here)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        res = _ScoreNetFn.apply(self, feats, names, need_grad, *params)
        return dict(zip(_OUT_KEYS, res))


def _diffuser_consts(model_conf, diffuser):
    """(coordinate_scaling, min_b, max_b, min_sigma, max_sigma, L) used by the score heads."""
    cs = float(model_conf.ipa.coordinate_scaling)
    r3 = getattr(diffuser, "_r3_diffuser", None)
    so3 = getattr(diffuser, "_so3_diffuser", None)
    min_b = float(getattr(r3, "min_b", 0.1))
    max_b = float(getattr(r3, "max_b", 20.0))
    min_s = float(getattr(so3, "min_sigma", 0.1))
    max_s = float(getattr(so3, "max_sigma", 1.5))
    cached = so3 if getattr(so3, "use_cached_score", False) else None   # icml_published.yaml: table lookup
    return (cs, min_b, max_b, min_s, max_s, 1000, cached, int(getattr(so3, "num_sigma", 1000)))
