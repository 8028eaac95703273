"""SE(3) diffuser -- drop-in for the reference's data/se3_diffuser.py (same class / method
signatures, return structures and error behaviour), engineered for MI355X:

* frames on a GPU never leave it: ``reverse`` / ``sample_ref`` / ``forward_marginal`` run the fp64
  kernels of csrc/fd_diffuser.hip on the frames' device and return quaternion-format Rigids, so the
  caller's ``to_tensor_7()`` is free (the reference goes device -> numpy -> scipy -> eigh -> device
  every reverse step, se3_diffuser.py:11-29,185,214; train_se3_diffusion.py:781);
* random draws come from the global numpy stream in the reference's call order (rotation noise
  first, then translation noise) unless the caller injects ``noise=`` / uses the ``*_device``
  fast entry points with a torch generator -- identical draws give identical frames;
* CPU-resident frames (forked DataLoader workers must not touch HIP: pdb_data_loader.py:253,260)
  take the same arithmetic in numpy.
"""
import logging

import numpy as np
import torch

from ..openfold.utils import rigid_utils as ru
from . import r3_diffuser, so3_diffuser
from . import utils as du


def _extract_trans_rots(rigid: ru.Rigid):
    """(trans [.., 3], rotvec [.., 3]) as numpy float64."""
    quat = rigid.get_rots().get_quats().detach().cpu().numpy().astype(np.float64)
    return rigid.get_trans().detach().cpu().numpy(), du.quat_wxyz_to_rotvec(quat)


def _assemble_rigid(rotvec, trans, device=None):
    quat = torch.tensor(du.rotvec_to_quat_wxyz(rotvec), dtype=torch.float32, device=device)
    return ru.Rigid(rots=ru.Rotation(quats=quat, normalize_quats=False),
                    trans=torch.tensor(np.asarray(trans), dtype=torch.float32, device=device))


def _f64(x, device):
    return torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(device=device, dtype=torch.float64).contiguous()


class SE3Diffuser:

    def __init__(self, se3_conf):
        self._log = logging.getLogger(__name__)
        self._se3_conf = se3_conf
        self._diffuse_rot = se3_conf.diffuse_rot
        self._so3_diffuser = so3_diffuser.SO3Diffuser(self._se3_conf.so3)
        self._diffuse_trans = se3_conf.diffuse_trans
        self._r3_diffuser = r3_diffuser.R3Diffuser(self._se3_conf.r3)

    # ------------------------------------------------------------------ forward process
    def forward_marginal(self, rigids_0: ru.Rigid, t: float, diffuse_mask: np.ndarray = None, as_tensor_7: bool = True):
        """Noise `rigids_0` ([.., N]) to time t.  Returns rigids_t (+ [.., N, 7] if as_tensor_7) and the DSM
        targets trans_score / rot_score (numpy float64) with their scalings."""
        if rigids_0.device.type == "cuda" and self._diffuse_rot and self._diffuse_trans:
            return self._forward_marginal_device(rigids_0, t, diffuse_mask, as_tensor_7)
        trans_0, rot_0 = _extract_trans_rots(rigids_0)
        if not self._diffuse_rot:
            rot_t, rot_score, rot_score_scaling = rot_0, np.zeros_like(rot_0), np.ones_like(t)
        else:
            rot_t, rot_score = self._so3_diffuser.forward_marginal(rot_0, t)
            rot_score_scaling = self._so3_diffuser.score_scaling(t)
        if not self._diffuse_trans:
            trans_t, trans_score, trans_score_scaling = trans_0, np.zeros_like(trans_0), np.ones_like(t)
        else:
            trans_t, trans_score = self._r3_diffuser.forward_marginal(trans_0, t)
            trans_score_scaling = self._r3_diffuser.score_scaling(t)
        if diffuse_mask is not None:
            m = diffuse_mask[..., None]
            rot_t = self._apply_mask(rot_t, rot_0, m)
            trans_t = self._apply_mask(trans_t, trans_0, m)
            trans_score = self._apply_mask(trans_score, np.zeros_like(trans_score), m)
            rot_score = self._apply_mask(rot_score, np.zeros_like(rot_score), m)
        rigids_t = _assemble_rigid(rot_t, trans_t, device=rigids_0.device)
        if as_tensor_7:
            rigids_t = rigids_t.to_tensor_7()
        return {'rigids_t': rigids_t, 'trans_score': trans_score, 'rot_score': rot_score,
                'trans_score_scaling': trans_score_scaling, 'rot_score_scaling': rot_score_scaling}

    def _forward_marginal_device(self, rigids_0, t, diffuse_mask, as_tensor_7):
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        from .. import hip
        dev = rigids_0.device
        shp = tuple(rigids_0.shape)
        n = int(np.prod(shp))
        z_axis, u = np.random.randn(n, 3), np.random.rand(n)           # so3 sample (rotation first)
        z_trans = np.random.normal(size=shp + (3,))                      # then r3
        so3, r3 = self._so3_diffuser, self._r3_diffuser
        cdf, omega = so3.device_tables(dev)
        idx = int(so3.t_to_idx(t))
        r0 = rigids_0.to_tensor_7().reshape(n, 7).contiguous()
        rt = torch.empty_like(r0)
        rs = torch.empty((n, 3), dtype=torch.float64, device=dev)
        ts = torch.empty((n, 3), dtype=torch.float64, device=dev)
        mask = None if diffuse_mask is None else torch.as_tensor(np.asarray(diffuse_mask), dtype=torch.float32, device=dev).reshape(n).contiguous()
        hip.get_lib().call("fd_forward_marginal", r0, _f64(z_axis, dev), _f64(u, dev), _f64(z_trans.reshape(n, 3), dev),
                           (cdf, idx * cdf.shape[1]), omega, omega.numel(), self._score_row(dev, idx),
                           float(so3.discrete_sigma[idx]), float(r3.marginal_b_t(t)), float(r3._r3_conf.coordinate_scaling), 1000, mask, rt, rs, ts, n)
        rt = rt.view(shp + (7,))
        out = rt if as_tensor_7 else ru.Rigid.from_tensor_7(rt)
        return {'rigids_t': out, 'trans_score': ts.view(shp + (3,)).cpu().numpy(), 'rot_score': rs.view(shp + (3,)).cpu().numpy(),
                'trans_score_scaling': r3.score_scaling(t), 'rot_score_scaling': so3.score_scaling(t)}

    def _score_row(self, dev, idx):
        """use_cached_score: (device score_norms table, element offset of sigma-bin idx); None = series in the kernel."""
        so3 = self._so3_diffuser
        if not so3.use_cached_score:
            return None
        tab = so3.device_score_norms(dev)
        return (tab, idx * tab.shape[1])

    def forward_marginal_batch(self, rigids_0, t, diffuse_mask=None, noise=None, generator=None):
        """Device-side training-batch generation: the reference noises every example in DataLoader workers
        with scipy (pdb_data_loader.py:240-262 -> forward_marginal per example); here a whole batch
        rigids_0 [B,N,7] (device) with per-example times t [B] is noised by one fd_forward_marginal_batch launch
        and everything stays on the device.  noise = (z_axis [B,N,3], u [B,N], z_trans [B,N,3]) float64 injects the draws
        (parity tests); by default they are drawn on the device in the reference's order (rotation axis, angle, then
        translation).  Returns the training-batch entries rigids_t [B,N,7] f32, rot_score / trans_score [B,N,3] f32,
        rot_score_scaling / trans_score_scaling [B] f32 (pdb_data_loader.py:245-262 casts them the same way)."""
        from .. import hip
        assert self._diffuse_rot and self._diffuse_trans, "device batch generation is built for diffuse_rot and diffuse_trans"
        dev = rigids_0.device
        B, N, _ = rigids_0.shape
        t = np.asarray(t, dtype=np.float64).reshape(B)
        so3, r3 = self._so3_diffuser, self._r3_diffuser
        cdf, omega = so3.device_tables(dev)
        if noise is None:
            z_axis = torch.randn((B, N, 3), dtype=torch.float64, device=dev, generator=generator)
            u = torch.rand((B, N), dtype=torch.float64, device=dev, generator=generator)
            z_trans = torch.randn((B, N, 3), dtype=torch.float64, device=dev, generator=generator)
        else:
            z_axis, u, z_trans = (_f64(x, dev).contiguous() for x in noise)
        r0 = rigids_0.to(torch.float32).contiguous()
        rt = torch.empty_like(r0)
        rs = torch.empty((B, N, 3), dtype=torch.float64, device=dev)
        ts = torch.empty((B, N, 3), dtype=torch.float64, device=dev)
        mask = None if diffuse_mask is None else torch.as_tensor(diffuse_mask, dtype=torch.float32, device=dev).contiguous()
        idx = np.asarray(so3.t_to_idx(t)).reshape(B)
        tparams = torch.tensor(np.stack([idx.astype(np.float64), so3.discrete_sigma[idx], r3.marginal_b_t(t)], 1),
                               dtype=torch.float64, device=dev)              # per example: sigma bin, sigma, marginal beta
        hip.get_lib().call("fd_forward_marginal_batch", r0, z_axis, u, z_trans, cdf, omega, omega.numel(),
                           so3.device_score_norms(dev) if so3.use_cached_score else None, tparams,
                           float(r3._r3_conf.coordinate_scaling), 1000, mask, rt, rs, ts, B, N)
        f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
        return {'rigids_t': rt, 'rot_score': rs.to(torch.float32), 'trans_score': ts.to(torch.float32),
                'rot_score_scaling': f32([so3.score_scaling(float(x)) for x in t]),
                'trans_score_scaling': f32([r3.score_scaling(float(x)) for x in t])}

    # ------------------------------------------------------------------ scores
    def calc_trans_0(self, trans_score, trans_t, t):
        return self._r3_diffuser.calc_trans_0(trans_score, trans_t, t)

    def calc_trans_score(self, trans_t, trans_0, t, use_torch=False, scale=True):
        return self._r3_diffuser.score(trans_t, trans_0, t, use_torch=use_torch, scale=scale)

    def calc_rot_score(self, rots_t, rots_0, t):
        """IGSO(3) score of rots_0^{-1} rots_t (Rotation objects, t [B]); differentiable w.r.t. rots_0."""
        quats_t, quats_0 = rots_t.get_quats(), rots_0.get_quats()
        if quats_t.is_cuda:
            from .. import score_ops
            return score_ops.rot_score(quats_t, quats_0, t, self)
        quats_0t = ru.quat_multiply(ru.invert_quat(quats_0), quats_t)
        return self._so3_diffuser.torch_score(du.quat_to_rotvec(quats_0t), t)

    def _apply_mask(self, x_diff, x_fixed, diff_mask):
        return diff_mask * x_diff + (1 - diff_mask) * x_fixed

    def trans_parameters(self, trans_t, score_t, t, dt, mask):
        # argument order kept as in the reference (se3_diffuser.py:130-132), which differs from
        # R3Diffuser.distribution(x_t, score_t, t, mask, dt)
        return self._r3_diffuser.distribution(trans_t, score_t, t, dt, mask)

    def score(self, rigid_0: ru.Rigid, rigid_t: ru.Rigid, t: float):
        tran_0, rot_0 = _extract_trans_rots(rigid_0)
        tran_t, rot_t = _extract_trans_rots(rigid_t)
        rot_score = np.zeros_like(rot_0) if not self._diffuse_rot else self._so3_diffuser.score(rot_t, t)
        trans_score = np.zeros_like(tran_0) if not self._diffuse_trans else self._r3_diffuser.score(tran_t, tran_0, t)
        return trans_score, rot_score

    def score_scaling(self, t):
        return self._so3_diffuser.score_scaling(t), self._r3_diffuser.score_scaling(t)

    # ------------------------------------------------------------------ reverse process
    def reverse(self, rigid_t: ru.Rigid, rot_score, trans_score, t: float, dt: float, diffuse_mask=None,
                center: bool = True, noise_scale: float = 1.0, noise=None):
        """One reverse-SDE step t -> t - dt.  rot_score / trans_score / diffuse_mask: numpy or tensors [.., N(,3)].
        noise=(z_rot, z_trans) injects standard-normal draws; default = the global numpy stream in the
        reference's order.  Returns a Rigid on rigid_t's device."""
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        shape3 = tuple(rigid_t.shape) + (3,)
        if noise is None:
            z_rot = np.random.normal(size=shape3) if self._diffuse_rot else np.zeros(shape3)
            z_trans = np.random.normal(size=shape3) if self._diffuse_trans else np.zeros(shape3)
        else:
            z_rot, z_trans = noise
        if rigid_t.device.type == "cuda":
            out7 = self.reverse_device(rigid_t.to_tensor_7(), rot_score, trans_score, t, dt, diffuse_mask=diffuse_mask,
                                       center=center, noise_scale=noise_scale, noise=(z_rot, z_trans))
            return ru.Rigid.from_tensor_7(out7)
        to_np = lambda x: x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
        rot_score, trans_score = to_np(rot_score), to_np(trans_score)
        trans_t, rot_t = _extract_trans_rots(rigid_t)
        rot_t_1 = rot_t if not self._diffuse_rot else self._so3_diffuser.reverse_with_noise(
            rot_t, rot_score, t, dt, noise_scale * to_np(z_rot))
        trans_t_1 = trans_t if not self._diffuse_trans else self._r3_diffuser.reverse_with_noise(
            trans_t, trans_score, t, dt, noise_scale * to_np(z_trans), center=center)
        if diffuse_mask is not None:
            m = to_np(diffuse_mask)[..., None]
            trans_t_1 = self._apply_mask(trans_t_1, trans_t, m)
            rot_t_1 = self._apply_mask(rot_t_1, rot_t, m)
        return _assemble_rigid(rot_t_1, trans_t_1, device=rigid_t.device)

    def reverse_device(self, rigids_t7, rot_score, trans_score, t, dt, diffuse_mask=None, center=True,
                       noise_scale=1.0, noise=None, generator=None, tparams=None, out=None):
        """Device-resident reverse step on [B, N, 7] frames (fd_se3_reverse_step).  noise=None draws
        z_rot then z_trans from `generator` (torch, on device).  tparams: optional device float64 [2] =
        (g_rot(t), b(t)) read by the kernel instead of the scalars (one captured hipGraph serves every t)."""
        from .. import hip
        dev = rigids_t7.device
        B, N = rigids_t7.shape[0], rigids_t7.shape[1]
        if noise is None:
            z_rot = torch.randn((B, N, 3), dtype=torch.float64, device=dev, generator=generator)
            z_trans = torch.randn((B, N, 3), dtype=torch.float64, device=dev, generator=generator)
        else:
            z_rot, z_trans = _f64(noise[0], dev), _f64(noise[1], dev)
        mask = None if diffuse_mask is None else torch.as_tensor(
            diffuse_mask.detach() if torch.is_tensor(diffuse_mask) else np.asarray(diffuse_mask)).to(device=dev, dtype=torch.float32).contiguous()
        if out is None:
            out = torch.empty((B, N, 7), dtype=torch.float32, device=dev)
        # float32 scores on the device (what the network returns) go in as they are: the kernel widens them in registers
        ok = lambda s_, dt_: torch.is_tensor(s_) and s_.dtype == dt_ and s_.device == dev and s_.is_contiguous()
        f32 = ok(rot_score, torch.float32) and ok(trans_score, torch.float32)
        net = ok(rot_score, torch.float64) and ok(trans_score, torch.float32)     # (as ScoreNetwork.forward returns them)
        scores = (rot_score, trans_score) if (f32 or net) else (_f64(rot_score, dev), _f64(trans_score, dev))
        entry = "fd_se3_reverse_step_f32" if f32 else "fd_se3_reverse_step_net" if net else "fd_se3_reverse_step"
        hip.get_lib().call(entry, rigids_t7.to(torch.float32).contiguous(),
                           scores[0], scores[1], z_rot, z_trans, mask, B, N, float(self._so3_diffuser.diffusion_coef(t)),
                           float(self._r3_diffuser.b_t(t)), tparams, float(dt), float(noise_scale),
                           float(self._r3_diffuser._r3_conf.coordinate_scaling), int(center), int(self._diffuse_rot),
                           int(self._diffuse_trans), out)
        return out

    # ------------------------------------------------------------------ prior
    def sample_ref(self, n_samples: int, impute: ru.Rigid = None, diffuse_mask: np.ndarray = None,
                   as_tensor_7: bool = False, device=None):
        """Samples rigids from the reference (prior) distribution: IGSO(3) at t=1 x N(0, I) (scaled units)."""
        if impute is not None:
            assert impute.shape[0] == n_samples
            trans_impute, rot_impute = _extract_trans_rots(impute)
            trans_impute = self._r3_diffuser._scale(trans_impute.reshape((n_samples, 3)))
            rot_impute = rot_impute.reshape((n_samples, 3))
        if diffuse_mask is not None and impute is None:
            raise ValueError('Must provide imputation values.')
        if (not self._diffuse_rot) and impute is None:
            raise ValueError('Must provide imputation values.')
        if (not self._diffuse_trans) and impute is None:
            raise ValueError('Must provide imputation values.')
        if device is not None and torch.device(device).type == "cuda" and impute is None:
            z_axis, u = np.random.randn(n_samples, 3), np.random.rand(n_samples)
            z_trans = np.random.normal(size=(n_samples, 3))
            t7 = self.sample_ref_device(n_samples, device, noise=(z_axis, u, z_trans))
            return {'rigids_t': t7 if as_tensor_7 else ru.Rigid.from_tensor_7(t7)}
        rot_ref = self._so3_diffuser.sample_ref(n_samples=n_samples) if self._diffuse_rot else rot_impute
        trans_ref = self._r3_diffuser.sample_ref(n_samples=n_samples) if self._diffuse_trans else trans_impute
        if diffuse_mask is not None:
            rot_ref = self._apply_mask(rot_ref, rot_impute, diffuse_mask[..., None])
            trans_ref = self._apply_mask(trans_ref, trans_impute, diffuse_mask[..., None])
        rigids_t = _assemble_rigid(rot_ref, self._r3_diffuser._unscale(trans_ref), device=device)
        if as_tensor_7:
            rigids_t = rigids_t.to_tensor_7()
        return {'rigids_t': rigids_t}

    def sample_ref_device(self, n_samples, device, noise=None, generator=None):
        """[n, 7] prior frames generated on `device` (fd_sample_ref)."""
        from .. import hip
        dev = torch.device(device)
        if noise is None:
            z_axis = torch.randn((n_samples, 3), dtype=torch.float64, device=dev, generator=generator)
            u = torch.rand((n_samples,), dtype=torch.float64, device=dev, generator=generator)
            z_trans = torch.randn((n_samples, 3), dtype=torch.float64, device=dev, generator=generator)
        else:
            z_axis, u, z_trans = (_f64(x, dev) for x in noise)
        so3 = self._so3_diffuser
        cdf, omega = so3.device_tables(dev)
        idx = int(so3.t_to_idx(1))
        out = torch.empty((n_samples, 7), dtype=torch.float32, device=dev)
        hip.get_lib().call("fd_sample_ref", z_axis, u, z_trans, (cdf, idx * cdf.shape[1]), omega, omega.numel(),
                           float(self._r3_diffuser._r3_conf.coordinate_scaling), out, n_samples)
        return out
