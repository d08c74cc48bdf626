"""Noise schedule, timesteps, eta(t) tables and the closed-form DDIM steps (CPU oracle; test
infrastructure).  All scalars are float64 numpy unless stated; tensors are torch CPU.

Follows (reference file:line):
  * scheduler construction           modules/inversion/diffusion_inversion.py:100-172,
                                     modules/models/__init__.py:134 (scaled_linear 0.00085..0.012,
                                     clip_sample=False, set_alpha_to_one=False, steps_offset=0)
  * inverse DDIM step ("sameshift")   modules/inverse_schedulers/scheduling_ddim_inverse.py:71-142
  * eta(t) table                      modules/inversion/eta_inversion.py:52-58, 107-139
  * backward DDIM-eta step / variance [3P] diffusers 0.21.1 DDIMScheduler.step/_get_variance as
                                     called at eta_inversion.py:245,310-312 (formulas: SURVEY App. B)
"""
import numpy as np
import torch

NUM_TRAIN = 1000


def alphas_cumprod(beta_start=0.00085, beta_end=0.012, n=NUM_TRAIN) -> np.ndarray:
    """fp32 cumprod exactly as the [3P] scheduler builds it (torch fp32 linspace**2, cumprod)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0).numpy().copy()


def timesteps_backward(S: int, steps_offset: int = 0) -> np.ndarray:
    """'leading' spacing: t_k = (S-1-k) * (1000 // S) + offset  (980..0 for S=50)."""
    ratio = NUM_TRAIN // S
    return (np.arange(0, S) * ratio).round()[::-1].astype(np.int64) + steps_offset


def timesteps_forward(S: int, steps_offset: int = 0) -> np.ndarray:
    """scheduling_ddim_inverse.py:51-69: reversed backward timesteps ("sameshift")."""
    return timesteps_backward(S, steps_offset)[::-1].copy()


def alpha_at(ac: np.ndarray, tau: int) -> float:
    """scheduling_ddim_inverse.py:85-92: clamp at 999; tau<0 -> final_alpha_cumprod = ac[0]."""
    tau = min(int(tau), NUM_TRAIN - 1)
    return float(ac[tau]) if tau >= 0 else float(ac[0])


def ddim_inverse_coeffs(ac: np.ndarray, t: int, S: int, inv_steps: str = "sameshift"):
    """Return (a_from, a_to) for the forward (inversion) step evaluated at loop timestep t
    (scheduling_ddim_inverse.py:127-137)."""
    d = NUM_TRAIN // S
    if inv_steps == "sameshift":
        t_from, t_to = t - d, t
    elif inv_steps in ("samesame", "shiftshift"):
        t_from, t_to = t, t + d
    else:
        raise Exception(inv_steps)
    return alpha_at(ac, t_from), alpha_at(ac, t_to)


def ddim_step(x: torch.Tensor, eps: torch.Tensor, a_from: float, a_to: float) -> torch.Tensor:
    """scheduling_ddim_inverse.py:94-98."""
    x0 = (x - (1 - a_from) ** 0.5 * eps) / a_from ** 0.5
    return a_to ** 0.5 * x0 + (1 - a_to) ** 0.5 * eps


def variance(ac: np.ndarray, t: int, S: int) -> float:
    """[3P] DDIMScheduler._get_variance(t, t - 1000//S) as called at eta_inversion.py:312."""
    p = t - NUM_TRAIN // S
    a_t = float(ac[t])
    a_p = float(ac[p]) if p >= 0 else float(ac[0])
    return (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)


def ddim_eta_step(x, eps, ac, t: int, S: int, eta, noise=None):
    """[3P] DDIMScheduler.step with (possibly per-pixel tensor) eta and explicit variance noise,
    as called at eta_inversion.py:245/260/310/369.  `eta` float or tensor broadcastable to x."""
    p = t - NUM_TRAIN // S
    a_t = float(ac[t])
    a_p = float(ac[p]) if p >= 0 else float(ac[0])
    var = (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)
    x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    std = eta * var ** 0.5
    direction = (1 - a_p - std ** 2) ** 0.5 * eps
    prev = a_p ** 0.5 * x0 + direction
    if noise is not None:
        prev = prev + std * noise
    return prev


def _eta_pow(p1, p2, p=1):
    """eta_inversion.py:52-58 without eval(): a*(clip(t,x1,x2)-x1)**p + y1."""
    (x1, y1), (x2, y2) = p1, p2
    a = (y2 - y1) / (x2 - x1) ** p
    return lambda t: a * (np.clip(t, x1, x2) - x1) ** p + y1


def eta_table(eta=(0.0, 0.4)) -> np.ndarray:
    """etas[1000] indexed by raw timestep (eta_inversion.py:121-139)."""
    if not isinstance(eta, (tuple, list)):
        eta = (eta, eta)
    if len(eta) == 3 or isinstance(eta[0], (tuple, list)):
        etas = _eta_pow(*eta)(np.linspace(0, 1, NUM_TRAIN))
    else:
        etas = np.linspace(eta[0], eta[1], NUM_TRAIN)
    return np.clip(etas, 0, None)


# ----------------------------------------------------------------------------- DPM-Solver++ (2M), forward and inverse
# [3P] diffusers 0.21.1 DPMSolverMultistepScheduler / DPMSolverMultistepInverseScheduler with the configuration the reference builds
# (diffusion_inversion.py:130-165: `from_config({**model.scheduler.config})` -> solver_order 2, algorithm_type "dpmsolver++", solver_type
# "midpoint", epsilon prediction, lower_order_final, no Karras sigmas, timestep_spacing "leading" inherited from the DDIM scheduler
# config of modules/models/__init__.py:134).  diffusers is not vendored in /root/reference and not installed: these functions restate the
# published scheduler from its documented formulas (DPM-Solver++ arXiv:2211.01095, eqs. 11-13) -- PARITY UNPINNED against diffusers
# itself; what IS checked: the first-order update equals the DDIM step exactly, the 2M update reproduces the exact solution for a
# noise prediction that is linear in lambda, and the native scheduler classes agree with these functions.
def dpm_tables(ac: np.ndarray):
    ac = np.asarray(ac, dtype=np.float64)
    alpha_t, sigma_t = np.sqrt(ac), np.sqrt(1.0 - ac)
    return alpha_t, sigma_t, np.log(alpha_t) - np.log(sigma_t)


def dpm_timesteps_backward(S: int, spacing: str = "leading", steps_offset: int = 0) -> np.ndarray:
    """DPMSolverMultistepScheduler.set_timesteps: S + 1 grid points over [0, 1000), reversed, the last one (0) dropped"""
    if spacing == "linspace":
        t = np.linspace(0, NUM_TRAIN - 1, S + 1).round()[::-1][:-1]
    else:
        t = (np.arange(0, S + 1) * (NUM_TRAIN // (S + 1))).round()[::-1][:-1] + steps_offset
    return t.astype(np.int64).copy()


def dpm_timesteps_forward(S: int, spacing: str = "leading", steps_offset: int = 0) -> np.ndarray:
    """DPMSolverMultistepInverseScheduler.set_timesteps: the same grid ascending from 0; the step after the last goes to the noisiest
    timestep 999"""
    if spacing == "linspace":
        t = np.linspace(0, NUM_TRAIN - 1, S + 1).round()[:-1]
    else:
        t = (np.arange(0, S + 1) * (NUM_TRAIN // (S + 1))).round()[:-1] + steps_offset
    return t.astype(np.int64).copy()


def dpm_x0(x, eps, tabs, t: int):
    """convert_model_output, epsilon prediction, dpmsolver++: the data prediction"""
    alpha_t, sigma_t, _ = tabs
    return (x - sigma_t[t] * eps) / alpha_t[t]


def dpm_first_order(x, m0, tabs, s: int, t: int):
    """x_t from x_s with data prediction m0 (DPM-Solver++ eq. 11); identical to a deterministic DDIM step s -> t"""
    alpha_t, sigma_t, lam = tabs
    h = lam[t] - lam[s]
    return (sigma_t[t] / sigma_t[s]) * x - (alpha_t[t] * (np.exp(-h) - 1.0)) * m0


def dpm_second_order(x, m0, m1, tabs, s1: int, s0: int, t: int):
    """multistep 2M midpoint update s0 -> t with the previous data prediction m1 taken at s1"""
    alpha_t, sigma_t, lam = tabs
    h, h0 = lam[t] - lam[s0], lam[s0] - lam[s1]
    r0 = h0 / h
    d1 = (1.0 / r0) * (m0 - m1)
    c = alpha_t[t] * (np.exp(-h) - 1.0)
    return (sigma_t[t] / sigma_t[s0]) * x - c * m0 - 0.5 * c * d1


class DpmStepper:
    """One direction of the multistep recursion with the scheduler's history (model_outputs, lower_order_nums).  `grid` = the loop
    timesteps, `last` = where the step after the last grid point lands (0 backward, 999 forward)."""

    def __init__(self, ac, grid, last: int, lower_order_final: bool = True):
        self.tabs, self.grid, self.last, self.lof = dpm_tables(ac), [int(t) for t in grid], int(last), lower_order_final
        self.prev_m, self.n_lower = None, 0

    def step(self, eps, t: int, x, i: int):
        """scheduler.step at loop index i (timestep t == grid[i])"""
        nxt = self.last if i == len(self.grid) - 1 else self.grid[i + 1]
        m0 = dpm_x0(x, eps, self.tabs, t)
        final = i == len(self.grid) - 1 and self.lof and len(self.grid) < 15
        if self.n_lower < 1 or final:
            out = dpm_first_order(x, m0, self.tabs, t, nxt)
        else:
            out = dpm_second_order(x, m0, self.prev_m, self.tabs, self.grid[i - 1], t, nxt)
        self.prev_m = m0
        self.n_lower = min(self.n_lower + 1, 2)
        return out


class DpmInverseStepper:
    """The reference's DPMSolverMultistepInverseScheduler.step (modules/inverse_schedulers/scheduling_dpmsolver_multistep_inverse.py:83-159)
    with its `inv_steps` modes, over the restated solver pieces above.  PINNED by tests/golden/dpm_inverse.npz (the reference's own class run
    over a restated [3P] solver, tests/golden/make_golden.py gen_dpm_inverse).
      samesame   the solver's ascending grid as is
      sameshift  the UNet is evaluated at grid[i] but the solver steps from grid[i-1] (first step: from the "first negative step"
                 grid[0] - (grid[1] - grid[0]), :73-80, :110-113)
      shiftshift the grid itself is shifted one step back (:57-63)
    Negative timesteps index the 1000-entry tables from the end and a step index of -1 / 0 makes `timesteps[step_index - 1]` wrap, exactly as
    the reference's tensor indexing does."""

    def __init__(self, ac, S: int, inv_steps: str = "samesame", spacing: str = "leading", lower_order_final: bool = True):
        assert inv_steps in ("samesame", "sameshift", "shiftshift")
        self.tabs, self.mode, self.lof = dpm_tables(ac), inv_steps, lower_order_final
        grid = [int(t) for t in dpm_timesteps_forward(S, spacing)]
        self.first_neg = grid[0] - (grid[1] - grid[0])                        # get_first_neg_step of the UNSHIFTED grid (:62)
        if inv_steps == "shiftshift":
            grid = [self.first_neg] + grid[:-1]
        self.grid, self.noisiest = grid, NUM_TRAIN - 1
        self.m, self.n_lower = [None, None], 0

    def step(self, eps, t: int, x):
        g, n = self.grid, len(self.grid)
        i = g.index(int(t)) if int(t) in g else n - 1
        ts = int(t)
        if self.mode == "sameshift":
            i -= 1
            ts = g[i] if i >= 0 else g[0] - (g[1] - g[0])                     # get_first_neg_step on the current grid (:73-80)
        nxt = self.noisiest if i == n - 1 else g[i + 1]
        final = i == n - 1 and self.lof and n < 15
        m0 = dpm_x0(x, eps, self.tabs, ts)
        self.m = [self.m[1], m0]
        if self.n_lower < 1 or final:
            out = dpm_first_order(x, m0, self.tabs, ts, nxt)
        else:
            out = dpm_second_order(x, m0, self.m[0], self.tabs, g[i - 1], ts, nxt)
        self.n_lower = min(self.n_lower + 1, 2)
        return out
