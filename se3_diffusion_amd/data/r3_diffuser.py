"""R^3 VP-SDE on translations -- drop-in for the reference's data/r3_diffuser.py (same class,
method names, argument meaning, ValueErrors).  These are the host-side scalar schedules and
numpy entry points; the batched device arithmetic lives in csrc/fd_diffuser.hip and is driven
by SE3Diffuser."""
import numpy as np
import torch


class R3Diffuser:
    """VP-SDE: dx = -1/2 b(t) x dt + sqrt(b(t)) dw,  b(t) = min_b + t (max_b - min_b)."""

    def __init__(self, r3_conf):
        self._r3_conf = r3_conf
        self.min_b = r3_conf.min_b
        self.max_b = r3_conf.max_b

    def _scale(self, x):
        return x * self._r3_conf.coordinate_scaling

    def _unscale(self, x):
        return x / self._r3_conf.coordinate_scaling

    def b_t(self, t):
        if np.any(t < 0) or np.any(t > 1):
            raise ValueError(f'Invalid t={t}')
        return self.min_b + t * (self.max_b - self.min_b)

    def diffusion_coef(self, t):
        return np.sqrt(self.b_t(t))

    def drift_coef(self, x, t):
        return -1 / 2 * self.b_t(t) * x

    def sample_ref(self, n_samples: float = 1):
        return np.random.normal(size=(n_samples, 3))

    def marginal_b_t(self, t):
        return t * self.min_b + (1 / 2) * (t ** 2) * (self.max_b - self.min_b)

    def calc_trans_0(self, score_t, x_t, t, use_torch=True):
        beta_t = self.marginal_b_t(t)[..., None, None]
        exp_fn = torch.exp if use_torch else np.exp
        return (score_t * (1 - exp_fn(-beta_t)) + x_t) / exp_fn(-1 / 2 * beta_t)

    def forward(self, x_t_1, t: float, num_t: int):
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        x_t_1 = self._scale(x_t_1)
        b_t = torch.tensor(self.marginal_b_t(t) / num_t).to(x_t_1.device)
        z = torch.tensor(np.random.normal(size=x_t_1.shape)).to(x_t_1.device)
        return torch.sqrt(1 - b_t) * x_t_1 + torch.sqrt(b_t) * z

    def distribution(self, x_t, score_t, t, mask, dt):
        x_t = self._scale(x_t)
        g_t = self.diffusion_coef(t)
        f_t = self.drift_coef(x_t, t)
        mu = x_t - (f_t - g_t ** 2 * score_t) * dt
        if mask is not None:
            mu *= mask[..., None]
        return mu, g_t * np.sqrt(dt)

    def forward_marginal(self, x_0: np.ndarray, t: float):
        """x_t ~ N(e^{-beta/2} x_0, 1 - e^{-beta}) in scaled units; returns (x_t in A, score)."""
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        x_0 = self._scale(x_0)
        beta = self.marginal_b_t(t)
        x_t = np.random.normal(loc=np.exp(-1 / 2 * beta) * x_0, scale=np.sqrt(1 - np.exp(-beta)))
        return self._unscale(x_t), self.score(x_t, x_0, t)

    def score_scaling(self, t: float):
        return 1 / np.sqrt(self.conditional_var(t))

    def reverse(self, *, x_t: np.ndarray, score_t: np.ndarray, t: float, dt: float, mask: np.ndarray = None,
                center: bool = True, noise_scale: float = 1.0):
        """One Euler-Maruyama step of the reverse SDE (host numpy entry point)."""
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        z = noise_scale * np.random.normal(size=score_t.shape)
        return self.reverse_with_noise(x_t, score_t, t, dt, z, mask=mask, center=center)

    def reverse_with_noise(self, x_t, score_t, t, dt, z, mask=None, center=True):
        x_t = self._scale(x_t)
        g_t = self.diffusion_coef(t)
        perturb = (self.drift_coef(x_t, t) - g_t ** 2 * score_t) * dt + g_t * np.sqrt(dt) * z
        if mask is not None:
            perturb = perturb * mask[..., None]
        else:
            mask = np.ones(x_t.shape[:-1])
        x_t_1 = x_t - perturb
        if center:
            com = np.sum(x_t_1, axis=-2) / np.sum(mask, axis=-1)[..., None]
            x_t_1 = x_t_1 - com[..., None, :]
        return self._unscale(x_t_1)

    def conditional_var(self, t, use_torch=False):
        if use_torch:
            return 1 - torch.exp(-self.marginal_b_t(t))
        return 1 - np.exp(-self.marginal_b_t(t))

    def score(self, x_t, x_0, t, use_torch=False, scale=False):
        exp_fn = torch.exp if use_torch else np.exp
        if scale:
            x_t = self._scale(x_t)
            x_0 = self._scale(x_0)
        return -(x_t - exp_fn(-1 / 2 * self.marginal_b_t(t)) * x_0) / self.conditional_var(t, use_torch=use_torch)
