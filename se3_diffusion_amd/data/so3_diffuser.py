"""IGSO(3) diffusion on rotations -- drop-in for the reference's data/so3_diffuser.py.

Same public surface (igso3_expansion, density, score, SO3Diffuser with sigma / t_to_idx /
diffusion_coef / sample / score / torch_score / score_scaling / forward_marginal / reverse,
the `_pdf/_cdf/_score_norms/_score_scaling` tables and the on-disk cache layout), different
engine:

* the [num_sigma x num_omega x L=1000] series behind the tables is evaluated by the HIP kernel
  ``fd_igso3_tables`` (fp64, rotation recurrence for sin/cos((l+1/2)w)) when a GPU is present --
  milliseconds instead of the reference's ~40-160 s of numpy;  GPU-less processes (forked
  DataLoader workers, CPU-only unit tests) evaluate the same recurrence in numpy, never
  materialising the [.., L] temporaries;
* ``torch_score`` on GPU tensors runs the fp64 series kernel of the score heads (fd_heads_fwd/bwd)
  and is differentiable, without the reference's device->host sync on ``t``.
"""
import logging
import os

import numpy as np
import torch

from . import utils as du


def _series(omega, eps, L=1000):
    """(f, f') of sum_l (2l+1) exp(-l(l+1) eps^2/2) sin((l+1/2) w)/sin(w/2); numpy float64, broadcast
    over omega/eps, O(L) passes with O(1) temporaries."""
    omega = np.asarray(omega, dtype=np.float64)
    eps = np.asarray(eps, dtype=np.float64)
    omega, eps = np.broadcast_arrays(omega, eps)
    lo, dlo = np.sin(omega / 2), 0.5 * np.cos(omega / 2)
    so, co = np.sin(omega), np.cos(omega)
    s, c = lo.copy(), 2 * dlo
    q = np.exp(-eps ** 2)
    w, r = np.ones_like(q), q.copy()
    f, df = np.zeros_like(q), np.zeros_like(q)
    for l in range(L):
        a = l + 0.5
        cw = (2 * l + 1) * w
        f += cw * s / lo
        df += cw * (lo * a * c - s * dlo) / lo ** 2
        w = w * r
        r = r * q
        if not np.any(w):
            break
        s, c = s * co + c * so, c * co - s * so
    return f, df


def _series_torch(omega, eps, L=1000):
    """(f, f') as _series, in torch float64 and differentiable w.r.t. omega (host tensors: the reference's torch_score is
    differentiable on the CPU too, data/so3_diffuser.py:274-305).  The sines / cosines of (l + 1/2) omega come from the same
    rotation recurrence; every step is an out-of-place torch op, so autograd sees the whole series."""
    omega = omega.to(torch.float64)
    eps = torch.as_tensor(eps, dtype=torch.float64)
    omega, eps = torch.broadcast_tensors(omega, eps)
    lo, dlo = torch.sin(omega / 2), 0.5 * torch.cos(omega / 2)
    so, co = torch.sin(omega), torch.cos(omega)
    s, c = lo, 2 * dlo
    q = torch.exp(-eps ** 2)
    w, r = torch.ones_like(q), q
    f, df = torch.zeros_like(q), torch.zeros_like(q)
    for l in range(L):
        cw = (2 * l + 1) * w
        f = f + cw * s / lo
        df = df + cw * (lo * (l + 0.5) * c - s * dlo) / lo ** 2
        w = w * r
        r = r * q
        if not bool(torch.any(w)):
            break
        s, c = s * co + c * so, c * co - s * so
    return f, df


def igso3_expansion(omega, eps, L=1000, use_torch=False):
    """Truncated IGSO(3) power series (eps = sqrt(2) * eps_leach); omega 1-D or 2-D."""
    if use_torch:
        if omega.dim() not in (1, 2):
            raise ValueError("Omega must be 1D or 2D.")
        f, _ = _series(omega.detach().cpu().numpy(), np.asarray(eps.detach().cpu().numpy() if torch.is_tensor(eps) else eps), L)
        return torch.as_tensor(f).to(omega.device)
    if np.ndim(omega) not in (1, 2):
        raise ValueError("Omega must be 1D or 2D.")
    return _series(omega, eps, L)[0]


def density(expansion, omega, marginal=True):
    """IGSO(3) density: over the rotation angle in [0, pi] (marginal) or over SO(3)."""
    if marginal:
        return expansion * (1 - np.cos(omega)) / np.pi
    return expansion / 8 / np.pi ** 2


def score(exp, omega, eps, L=1000, use_torch=False):
    """d/d omega log IGSO3(omega; eps): f'(omega) / (f(omega) + 1e-4) by the quotient rule."""
    if use_torch:
        _, df = _series(omega.detach().cpu().numpy(), np.asarray(eps.detach().cpu().numpy() if torch.is_tensor(eps) else eps), L)
        return torch.as_tensor(df).to(omega.device) / (exp + 1e-4)
    if np.ndim(omega) > 2:
        raise ValueError("Omega must be 1D or 2D.")
    return _series(omega, eps, L)[1] / (exp + 1e-4)


def _build_tables(discrete_sigma, discrete_omega, L=1000):
    """pdf, cdf, score_norms [num_sigma, num_omega] float64."""
    ns, no = len(discrete_sigma), len(discrete_omega)
    if torch.cuda.is_available():
        from .. import hip
        dev = torch.device("cuda", torch.cuda.current_device())
        sg = torch.tensor(discrete_sigma, dtype=torch.float64, device=dev)
        om = torch.tensor(discrete_omega, dtype=torch.float64, device=dev)
        pdf = torch.empty((ns, no), dtype=torch.float64, device=dev)
        cdf = torch.empty_like(pdf)
        sn = torch.empty_like(pdf)
        hip.get_lib().call("fd_igso3_tables", sg, om, ns, no, L, pdf, cdf, sn)
        return pdf.cpu().numpy(), cdf.cpu().numpy(), sn.cpu().numpy()
    f, df = _series(discrete_omega[None, :], np.asarray(discrete_sigma)[:, None], L)
    pdf = density(f, discrete_omega[None, :], marginal=True)
    cdf = pdf.cumsum(axis=-1) / no * np.pi
    return pdf, cdf, df / (f + 1e-4)


class SO3Diffuser:

    def __init__(self, so3_conf):
        self.schedule = so3_conf.schedule
        self.min_sigma = so3_conf.min_sigma
        self.max_sigma = so3_conf.max_sigma
        self.num_sigma = so3_conf.num_sigma
        self.use_cached_score = so3_conf.use_cached_score
        self._log = logging.getLogger(__name__)
        self.discrete_omega = np.linspace(0, np.pi, so3_conf.num_omega + 1)[1:]   # skip omega = 0

        dots = lambda x: str(x).replace('.', '_')
        cache_dir = os.path.join(
            so3_conf.cache_dir,
            f'eps_{so3_conf.num_sigma}_omega_{so3_conf.num_omega}_min_sigma_{dots(so3_conf.min_sigma)}'
            f'_max_sigma_{dots(so3_conf.max_sigma)}_schedule_{so3_conf.schedule}')
        os.makedirs(cache_dir, exist_ok=True)
        paths = [os.path.join(cache_dir, n) for n in ('pdf_vals.npy', 'cdf_vals.npy', 'score_norms.npy')]
        if all(os.path.exists(p) for p in paths):
            self._log.info(f'Using cached IGSO3 in {cache_dir}')
            self._pdf, self._cdf, self._score_norms = (np.load(p) for p in paths)
        else:
            self._log.info(f'Computing IGSO3. Saving in {cache_dir}')
            self._pdf, self._cdf, self._score_norms = _build_tables(self.discrete_sigma, self.discrete_omega)
            for p, a in zip(paths, (self._pdf, self._cdf, self._score_norms)):
                np.save(p, a)
        self._score_scaling = np.sqrt(np.abs(
            np.sum(self._score_norms ** 2 * self._pdf, axis=-1) / np.sum(self._pdf, axis=-1))) / np.sqrt(3)
        self._dev_tables = {}

    @property
    def discrete_sigma(self):
        return self.sigma(np.linspace(0.0, 1.0, self.num_sigma))

    def sigma_idx(self, sigma: np.ndarray):
        return np.digitize(sigma, self.discrete_sigma) - 1

    def sigma(self, t: np.ndarray):
        if np.any(t < 0) or np.any(t > 1):
            raise ValueError(f'Invalid t={t}')
        if self.schedule != 'logarithmic':
            raise ValueError(f'Unrecognize schedule {self.schedule}')
        return np.log(t * np.exp(self.max_sigma) + (1 - t) * np.exp(self.min_sigma))

    def diffusion_coef(self, t):
        if self.schedule != 'logarithmic':
            raise ValueError(f'Unrecognize schedule {self.schedule}')
        s = self.sigma(t)
        return np.sqrt(2 * (np.exp(self.max_sigma) - np.exp(self.min_sigma)) * s / np.exp(s))

    def t_to_idx(self, t: np.ndarray):
        return self.sigma_idx(self.sigma(t))

    def device_tables(self, device):
        """(cdf [ns, no], omega grid [no]) float64 on `device` for the sampling kernels."""
        key = str(device)
        if key not in self._dev_tables:
            self._dev_tables[key] = (torch.tensor(self._cdf, dtype=torch.float64, device=device),
                                     torch.tensor(self.discrete_omega, dtype=torch.float64, device=device))
        return self._dev_tables[key]

    def device_score_norms(self, device):
        """score_norms [ns, no] float64 on `device` (the use_cached_score lookup table of the score kernels)."""
        key = ("score_norms", str(device))
        if key not in self._dev_tables:
            self._dev_tables[key] = torch.tensor(self._score_norms, dtype=torch.float64, device=device)
        return self._dev_tables[key]

    def sample_igso3(self, t: float, n_samples: float = 1):
        """Inverse-CDF sample of the rotation angle at time t."""
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        x = np.random.rand(n_samples)
        return np.interp(x, self._cdf[self.t_to_idx(t)], self.discrete_omega)

    def sample(self, t: float, n_samples: float = 1):
        """[n, 3] rotation vectors ~ IGSO(3): uniform axis (normalised Gaussian) x sampled angle."""
        x = np.random.randn(n_samples, 3)
        x /= np.linalg.norm(x, axis=-1, keepdims=True)
        return x * self.sample_igso3(t, n_samples=n_samples)[:, None]

    def sample_ref(self, n_samples: float = 1):
        return self.sample(1, n_samples=n_samples)

    def score(self, vec: np.ndarray, t: float, eps: float = 1e-6):
        """Score of IGSO(3) at rotation vectors `vec` (numpy, float64)."""
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        vec = np.asarray(vec)
        omega = np.linalg.norm(vec, axis=-1) + eps
        if self.use_cached_score:
            row = self._score_norms[self.t_to_idx(t)]
            idx = np.searchsorted(self.discrete_omega[:-1], omega, side='left')
            scal = row[idx]
        else:
            sg = self.discrete_sigma[self.t_to_idx(t)]
            f, df = _series(omega, sg)
            scal = df / (f + 1e-4)
        return scal[..., None] * vec / (omega[..., None] + eps)

    def torch_score(self, vec: torch.Tensor, t: torch.Tensor, eps: float = 1e-6):
        """[B, N, 3] rotation vectors, t [B] -> float64 score.  GPU tensors: fp64 HIP series, differentiable; host tensors:
        numpy series, or torch float64 (differentiable) when `vec` requires a gradient."""
        if vec.is_cuda:
            from .. import score_ops
            return score_ops.rotvec_score(vec, t, self)
        if vec.requires_grad:
            # host tensors that carry a gradient: the same arithmetic in torch float64 (autograd through the series; with
            # use_cached_score the tabulated magnitude is piecewise constant in omega, as in the reference's bucketize + gather)
            tt = np.atleast_1d(du.move_to_np(t)).astype(np.float64)
            omega = torch.linalg.norm(vec.to(torch.float64), dim=-1) + eps
            sg = torch.as_tensor(self.discrete_sigma[self.t_to_idx(tt)], dtype=torch.float64)
            sg = sg.reshape(sg.shape + (1,) * (omega.dim() - 1))
            if self.use_cached_score:
                rows = torch.as_tensor(self._score_norms[self.t_to_idx(tt)], dtype=torch.float64)
                idx = torch.bucketize(omega.detach(), torch.as_tensor(self.discrete_omega[:-1], dtype=torch.float64))
                scal = torch.gather(rows, 1, idx.reshape(rows.shape[0], -1)).reshape(omega.shape)
            else:
                f, df = _series_torch(omega, sg)
                scal = df / (f + 1e-4)
            return scal[..., None] * vec.to(torch.float64) / (omega[..., None] + eps)
        tt = np.atleast_1d(du.move_to_np(t)).astype(np.float64)
        v = du.move_to_np(vec)
        omega = np.linalg.norm(v, axis=-1) + eps
        sg = self.discrete_sigma[self.t_to_idx(tt)]
        sg = sg.reshape(sg.shape + (1,) * (omega.ndim - 1))
        if self.use_cached_score:
            rows = self._score_norms[self.t_to_idx(tt)]
            idx = np.searchsorted(self.discrete_omega[:-1], omega, side='left')
            scal = np.take_along_axis(rows, idx.reshape(rows.shape[0], -1), axis=1).reshape(omega.shape)
        else:
            f, df = _series(omega.astype(np.float64), sg)
            scal = df / (f + 1e-4)
        return torch.as_tensor(scal[..., None] * v / (omega[..., None] + eps))

    def score_scaling(self, t: np.ndarray):
        return self._score_scaling[self.t_to_idx(t)]

    def forward_marginal(self, rot_0: np.ndarray, t: float):
        """rot_t = rot_0 o sample (right multiplication); score of the sampled rotation."""
        n_samples = np.cumprod(rot_0.shape[:-1])[-1]
        sampled = self.sample(t, n_samples=n_samples)
        rot_score = self.score(sampled, t).reshape(rot_0.shape)
        rot_t = du.compose_rotvec(rot_0.reshape(-1, 3), sampled).reshape(rot_0.shape)
        return rot_t, rot_score

    def reverse(self, rot_t: np.ndarray, score_t: np.ndarray, t: float, dt: float, mask: np.ndarray = None,
                noise_scale: float = 1.0):
        """One geodesic-random-walk step of the reverse SDE (host numpy entry point)."""
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        z = noise_scale * np.random.normal(size=score_t.shape)
        return self.reverse_with_noise(rot_t, score_t, t, dt, z, mask=mask)

    def reverse_with_noise(self, rot_t, score_t, t, dt, z, mask=None):
        g_t = self.diffusion_coef(t)
        perturb = (g_t ** 2) * score_t * dt + g_t * np.sqrt(dt) * z
        if mask is not None:
            perturb = perturb * mask[..., None]
        return du.compose_rotvec(rot_t.reshape(-1, 3), perturb.reshape(-1, 3)).reshape(rot_t.shape)
