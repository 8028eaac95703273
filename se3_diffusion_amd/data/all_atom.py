"""Idealised backbone from frames -- the hot-path part of the reference's data/all_atom.py.

``compute_backbone(bb_rigids, psi_torsions)`` keeps the reference signature and return tuple
(atom37 [.., 37, 3], atom37_mask, aatype, atom14 [.., 14, 3]); the atoms come from one HIP kernel
(fd_backbone_atoms) on the frames' device, with no CPU-resident index tensors and therefore none of
the reference's device<->host copies per call (all_atom.py:157-172).  Forward only: inside
ScoreNetwork the differentiable version is fused into the score-head kernels."""
import torch

from ..openfold.utils import rigid_utils as ru
from .. import train_step as _ts

Rigid = ru.Rigid
Rotation = ru.Rotation


def compute_backbone(bb_rigids, psi_torsions):
    rig7 = bb_rigids.to_tensor_7().detach()
    atom37, atom14 = _ts.backbone_atoms(rig7, psi_torsions.detach().to(rig7.device))
    atom37_mask = torch.any(atom37 != 0, dim=-1)
    aatype = torch.zeros(bb_rigids.shape, dtype=torch.long)
    return atom37, atom37_mask, aatype, atom14
