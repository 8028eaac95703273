"""Translation half of the SE(3) diffuser: a variance-preserving SDE on R^3.

Drop-in for the reference's ``data/r3_diffuser.py`` (class name, method names, argument meaning and the
``ValueError`` on a bad / non-scalar ``t`` are the boundary contract).  Only host-side scalar
schedules and the numpy entry points live here -- forked DataLoader workers call ``forward_marginal`` /
``sample_ref`` and must not touch HIP; the batched device arithmetic is ``csrc/fd_diffuser.hip``, driven by
``SE3Diffuser``.

The process, written once (everything below is derived from these four closed forms):

    dx = -1/2 beta(t) x dt + sqrt(beta(t)) dw            beta(t)  = min_b + t (max_b - min_b)      [b_t]
    B(t) = int_0^t beta = t min_b + t^2 (max_b - min_b)/2                                           [marginal_b_t]
    x_t | x_0 ~ N(a(t) x_0, v(t) I)                      a(t) = exp(-B(t)/2),  v(t) = 1 - exp(-B(t))
    grad log p(x_t | x_0) = -(x_t - a(t) x_0) / v(t)

Coordinates enter in Angstrom and are diffused in ``coordinate_scaling`` units (0.1: nanometres).
"""
import numpy as np
import torch


def _require_scalar(t):
    if not np.isscalar(t):
        raise ValueError(f'{t} must be a scalar.')


class R3Diffuser:

    def __init__(self, r3_conf):
        self._r3_conf = r3_conf
        self.min_b = r3_conf.min_b
        self.max_b = r3_conf.max_b

    # ---- units ------------------------------------------------------------------------------------------------
    def _scale(self, x):
        return x * self._r3_conf.coordinate_scaling

    def _unscale(self, x):
        return x / self._r3_conf.coordinate_scaling

    # ---- schedule ---------------------------------------------------------------------------------------------
    def b_t(self, t):
        """beta(t); t outside [0, 1] is an error (reference r3_diffuser.py:26-29)."""
        if np.any(t < 0) or np.any(t > 1):
            raise ValueError(f'Invalid t={t}')
        return self.min_b + t * (self.max_b - self.min_b)

    def marginal_b_t(self, t):
        """B(t), the integrated rate (works on floats, numpy arrays and torch tensors alike)."""
        return t * self.min_b + (1 / 2) * (t ** 2) * (self.max_b - self.min_b)

    def _mean_coef(self, t, use_torch=False):
        return (torch.exp if use_torch else np.exp)(-1 / 2 * self.marginal_b_t(t))

    def conditional_var(self, t, use_torch=False):
        """v(t)"""
        return 1 - (torch.exp if use_torch else np.exp)(-self.marginal_b_t(t))

    def diffusion_coef(self, t):
        return np.sqrt(self.b_t(t))

    def drift_coef(self, x, t):
        return -1 / 2 * self.b_t(t) * x

    def score_scaling(self, t: float):
        """1 / sqrt(v(t)): the loss weights the translation score error by v(t)."""
        return 1 / np.sqrt(self.conditional_var(t))

    # ---- densities --------------------------------------------------------------------------------------------
    def score(self, x_t, x_0, t, use_torch=False, scale=False):
        """grad_{x_t} log p(x_t | x_0); ``scale`` converts Angstrom inputs first."""
        if scale:
            x_t, x_0 = self._scale(x_t), self._scale(x_0)
        return -(x_t - self._mean_coef(t, use_torch) * x_0) / self.conditional_var(t, use_torch=use_torch)

    def calc_trans_0(self, score_t, x_t, t, use_torch=True):
        """Solve the score identity for x_0 (Tweedie); t is a per-example vector [B]."""
        big_b = self.marginal_b_t(t)[..., None, None]
        e = torch.exp if use_torch else np.exp
        return (score_t * (1 - e(-big_b)) + x_t) / e(-1 / 2 * big_b)

    # ---- sampling (numpy, global RNG: the reference's draw order is part of the contract) ----------------------
    def sample_ref(self, n_samples: float = 1):
        """Prior: unit Gaussian in scaled units."""
        return np.random.normal(size=(n_samples, 3))

    def forward_marginal(self, x_0: np.ndarray, t: float):
        """Draw x_t | x_0 in one shot.  Returns (x_t in Angstrom, its conditional score in scaled units)."""
        _require_scalar(t)
        x_0 = self._scale(x_0)
        x_t = np.random.normal(loc=self._mean_coef(t) * x_0, scale=np.sqrt(self.conditional_var(t)))
        return self._unscale(x_t), self.score(x_t, x_0, t)

    def forward(self, x_t_1, t: float, num_t: int):
        """One discretised forward step x_{t-1} -> x_t with per-step rate B(t) / num_t (unused by the callers;
        kept because it is public in the reference, r3_diffuser.py:52-69)."""
        _require_scalar(t)
        x = self._scale(x_t_1)
        rate = torch.tensor(self.marginal_b_t(t) / num_t).to(x.device)
        noise = torch.tensor(np.random.normal(size=x.shape)).to(x.device)
        return torch.sqrt(1 - rate) * x + torch.sqrt(rate) * noise

    # ---- reverse SDE -------------------------------------------------------------------------------------------
    def _reverse_drift(self, x_scaled, score_t, t):
        """f(x, t) - g(t)^2 score: the drift of the time-reversed SDE (before the sign of -dt)."""
        return self.drift_coef(x_scaled, t) - self.b_t(t) * score_t

    def distribution(self, x_t, score_t, t, mask, dt):
        """(mean, std) of one Euler-Maruyama reverse step in scaled units (public in the reference :71-79, unused)."""
        x = self._scale(x_t)
        mean = x - self._reverse_drift(x, score_t, t) * dt
        if mask is not None:
            mean *= mask[..., None]
        return mean, self.diffusion_coef(t) * np.sqrt(dt)

    def reverse(self, *, x_t: np.ndarray, score_t: np.ndarray, t: float, dt: float, mask: np.ndarray = None,
                center: bool = True, noise_scale: float = 1.0):
        """One Euler-Maruyama step of the reverse SDE on host arrays (Angstrom in, Angstrom out); the noise is
        drawn here from numpy's global stream, as the reference does (r3_diffuser.py:134)."""
        _require_scalar(t)
        noise = noise_scale * np.random.normal(size=score_t.shape)
        return self.reverse_with_noise(x_t, score_t, t, dt, noise, mask=mask, center=center)

    def reverse_with_noise(self, x_t, score_t, t, dt, z, mask=None, center=True):
        """The same step with the (already scaled) noise ``z`` supplied: what fd_se3_reverse_step computes on the
        device.  Masked-out residues do not move; the centre of mass is taken over sum(mask) residues."""
        x = self._scale(x_t)
        move = self._reverse_drift(x, score_t, t) * dt + self.diffusion_coef(t) * np.sqrt(dt) * z
        if mask is None:
            mask = np.ones(x.shape[:-1])
        else:
            move = move * mask[..., None]
        x_next = x - move
        if center:
            x_next = x_next - (np.sum(x_next, axis=-2) / np.sum(mask, axis=-1)[..., None])[..., None, :]
        return self._unscale(x_next)
