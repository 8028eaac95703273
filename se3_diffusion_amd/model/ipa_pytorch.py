"""Parameter containers of the FrameDiff trunk (drop-in for model/ipa_pytorch.py).

Same class names, constructor signatures, attribute names and therefore the same
``state_dict`` keys / shapes as the reference -- but these modules only
HOLD parameters: all arithmetic of IpaScore.forward (reference ipa_pytorch.py:611-672)
runs in the HIP kernels driven by se3_diffusion_amd/trunk.py.
"""
import math

import torch
from torch import nn

_TRUNC_STD = 0.87962566103423978  # std of a unit normal truncated to [-2, 2]


def _trunc_normal_(w, scale):
    fan_in = w.shape[1]
    std = math.sqrt(scale / max(1, fan_in)) / _TRUNC_STD
    with torch.no_grad():
        nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2.0 * std, b=2.0 * std)


class Linear(nn.Linear):
    """nn.Linear with the AF2 initialisers the reference selects by name
    (ipa_pytorch.py:101-166): default = LeCun trunc-normal, relu = He trunc-normal,
    glorot, gating (w=0,b=1), normal, final (zeros).  Bias starts at 0."""

    def __init__(self, in_dim, out_dim, bias=True, init="default", init_fn=None):
        super().__init__(in_dim, out_dim, bias=bias)
        with torch.no_grad():
            if bias:
                self.bias.zero_()
            if init_fn is not None:
                init_fn(self.weight, self.bias)
            elif init == "default":
                _trunc_normal_(self.weight, 1.0)
            elif init == "relu":
                _trunc_normal_(self.weight, 2.0)
            elif init == "glorot":
                nn.init.xavier_uniform_(self.weight, gain=1)
            elif init == "gating":
                self.weight.zero_()
                if bias:
                    self.bias.fill_(1.0)
            elif init == "normal":
                nn.init.kaiming_normal_(self.weight, nonlinearity="linear")
            elif init == "final":
                self.weight.zero_()
            else:
                raise ValueError("Invalid init string.")

    def forward(self, x):  # pragma: no cover - containers are not executed
        raise RuntimeError("parameter container; the computation runs in the fused HIP path (ScoreNetwork.forward)")


class StructureModuleTransition(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c = c
        self.linear_1 = Linear(c, c, init="relu")
        self.linear_2 = Linear(c, c, init="relu")
        self.linear_3 = Linear(c, c, init="final")
        self.ln = nn.LayerNorm(c)


class EdgeTransition(nn.Module):
    def __init__(self, *, node_embed_size, edge_embed_in, edge_embed_out, num_layers=2, node_dilation=2):
        super().__init__()
        bias_embed_size = node_embed_size // node_dilation
        self.initial_embed = Linear(node_embed_size, bias_embed_size, init="relu")
        hidden = bias_embed_size * 2 + edge_embed_in
        layers = []
        for _ in range(num_layers):
            layers += [Linear(hidden, hidden, init="relu"), nn.ReLU()]
        self.trunk = nn.Sequential(*layers)
        self.final_layer = Linear(hidden, edge_embed_out, init="final")
        self.layer_norm = nn.LayerNorm(edge_embed_out)


class InvariantPointAttention(nn.Module):
    def __init__(self, ipa_conf, inf=1e5, eps=1e-8):
        super().__init__()
        self._ipa_conf = ipa_conf
        c_s, c_z, c_h = ipa_conf.c_s, ipa_conf.c_z, ipa_conf.c_hidden
        nh, pq, pv = ipa_conf.no_heads, ipa_conf.no_qk_points, ipa_conf.no_v_points
        self.linear_q = Linear(c_s, nh * c_h)
        self.linear_kv = Linear(c_s, 2 * nh * c_h)
        self.linear_q_points = Linear(c_s, nh * pq * 3)
        self.linear_kv_points = Linear(c_s, nh * (pq + pv) * 3)
        self.linear_b = Linear(c_z, nh)
        self.down_z = Linear(c_z, c_z // 4)
        self.head_weights = nn.Parameter(torch.full((nh,), 0.541324854612918))  # softplus^-1(1)
        self.linear_out = Linear(nh * (c_z // 4 + c_h + pv * 4), c_s, init="final")
        self.linear_rbf = Linear(20, 1)  # unused; present in published checkpoints


class TorsionAngles(nn.Module):
    def __init__(self, c, num_torsions, eps=1e-8):
        super().__init__()
        self.linear_1 = Linear(c, c, init="relu")
        self.linear_2 = Linear(c, c, init="relu")
        self.linear_3 = Linear(c, c, init="final")  # unused; present in published checkpoints
        self.linear_final = Linear(c, num_torsions * 2, init="final")


class BackboneUpdate(nn.Module):
    def __init__(self, c_s):
        super().__init__()
        self.linear = Linear(c_s, 6, init="final")


class IpaScore(nn.Module):
    def __init__(self, model_conf, diffuser):
        super().__init__()
        self._model_conf = model_conf
        ipa = model_conf.ipa
        self._ipa_conf = ipa
        self.diffuser = diffuser
        self.trunk = nn.ModuleDict()
        d = ipa.c_s + ipa.c_skip
        for b in range(ipa.num_blocks):
            self.trunk[f"ipa_{b}"] = InvariantPointAttention(ipa)
            self.trunk[f"ipa_ln_{b}"] = nn.LayerNorm(ipa.c_s)
            self.trunk[f"skip_embed_{b}"] = Linear(model_conf.node_embed_size, ipa.c_skip, init="final")
            layer = nn.TransformerEncoderLayer(d_model=d, nhead=ipa.seq_tfmr_num_heads, dim_feedforward=d,
                                               batch_first=True, dropout=0.0, norm_first=False)
            self.trunk[f"seq_tfmr_{b}"] = nn.TransformerEncoder(layer, ipa.seq_tfmr_num_layers)
            self.trunk[f"post_tfmr_{b}"] = Linear(d, ipa.c_s, init="final")
            self.trunk[f"node_transition_{b}"] = StructureModuleTransition(c=ipa.c_s)
            self.trunk[f"bb_update_{b}"] = BackboneUpdate(ipa.c_s)
            if b < ipa.num_blocks - 1:
                self.trunk[f"edge_transition_{b}"] = EdgeTransition(
                    node_embed_size=ipa.c_s, edge_embed_in=model_conf.edge_embed_size,
                    edge_embed_out=model_conf.edge_embed_size)
        self.torsion_pred = TorsionAngles(ipa.c_s, 1)
