"""CPU oracle for the diffusion glue -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates src/thor/pipelines.py (noise process, loss, predictor/corrector
sampler) and src/thor/score.py (sliding-window score function + Gaussian
likelihood guidance) with explicit randomness so tests can inject the draws.
Pinned by tests/golden/{kat,sampler}_*.npz (generated from the imported
reference by tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional

import torch
from torch import Tensor


# ---------------------------------------------------------------- noise process
def alpha(t: Tensor, eta: float = 1e-3) -> Tensor:
    """src/thor/pipelines.py:13-14."""
    return torch.cos(math.acos(math.sqrt(eta)) * t) ** 2


def mu(t: Tensor, eta: float = 1e-3) -> Tensor:
    """src/thor/pipelines.py:16-17."""
    return alpha(t, eta)


def sigma(t: Tensor, eta: float = 1e-3) -> Tensor:
    """src/thor/pipelines.py:19-20."""
    return (1 - alpha(t, eta) ** 2 + eta**2).sqrt()


def perturb(x: Tensor, t: Tensor, eps: Tensor, eta: float = 1e-3) -> Tensor:
    """src/thor/pipelines.py:22-25 with the normal draw ``eps`` injected."""
    return mu(t, eta) * x + sigma(t, eta) * eps


def loss(net: Callable, x: Tensor, t: Tensor, eps: Tensor, eta: float = 1e-3) -> Tensor:
    """src/thor/pipelines.py:27-35 with ``t ~ U(0,1)`` of shape (B,1,1,1) and ``eps`` injected.
    Unreduced squared error; the caller takes ``.mean()`` (training_loop.py:377)."""
    return (net(perturb(x, t, eps, eta), t) - eps) ** 2


# ---------------------------------------------------------------- score functions
def unfold_windows(x: Tensor, k: int) -> Tensor:
    """src/thor/score.py:68-74: (L,C,H,W) -> (L-w+1, w*C, H, W), channel = tau*C + c."""
    w = 2 * k + 1
    L = x.shape[0]
    return torch.stack([x[i : i + w].reshape(-1, *x.shape[2:]) for i in range(L - w + 1)], dim=0)


def fold_windows(y: Tensor, k: int) -> Tensor:
    """src/thor/score.py:76-88: first window's leading k frames, every centre, last window's trailing k."""
    w = 2 * k + 1
    y = y.reshape(y.shape[0], w, -1, *y.shape[2:])
    return torch.cat((y[0, :k], y[:, k], y[-1, k + 1 :]), dim=0)


def window_score(net: Callable, x: Tensor, t: Tensor, k: int, batch_size: Optional[int] = None) -> Tensor:
    """DefaultScoreFunction.score_fn (src/thor/score.py:90-93) when ``batch_size`` is None, else
    BatchedScoreFunction.score_fn (src/thor/score.py:156-185) -- same result, windows fed in chunks."""
    win = unfold_windows(x, k)
    if batch_size is None:
        return fold_windows(net(win, t), k)
    outs = [net(chunk, t) for chunk in win.split(batch_size, 0)]
    return fold_windows(torch.cat(outs, 0), k)


class GuidedScore:
    """AbstractScoreFunction.__call__/condition_on (src/thor/score.py:24-60).

    ``eps - sigma * d/dx log p(y | x0_hat(x))`` with
    ``log p = -1/2 sum (y - A(x0_hat))^2 / (std^2 + gamma (sigma/mu)^2)``.
    ``exact_grad=False`` treats the network output as a constant of the derivative.
    Uses plain autograd instead of ``torch.func.jacrev`` (a scalar's Jacobian is its gradient).
    """

    def __init__(self, net, k, A=None, y=None, std=None, gamma=1e-2, exact_grad=True, batch_size=None, eta=1e-3):
        self.net, self.k, self.A, self.y, self.std = net, k, A, y, std
        self.gamma, self.exact_grad, self.batch_size, self.eta = gamma, exact_grad, batch_size, eta

    def __call__(self, x: Tensor, t: Tensor) -> Tensor:
        if self.A is None:
            return window_score(self.net, x, t, self.k, self.batch_size)
        m, s = mu(t, self.eta), sigma(t, self.eta)
        with torch.enable_grad():
            xg = x.detach().requires_grad_(True)
            with torch.set_grad_enabled(self.exact_grad):
                eps = window_score(self.net, xg, t, self.k, self.batch_size)
            x0 = (xg - s * eps) / m
            err = self.y - self.A(x0)
            var = self.std**2 + self.gamma * (s / m) ** 2
            logp = -(err**2 / var).sum() / 2
            (J,) = torch.autograd.grad(logp, xg)
        return eps.detach() - s * J


# ---------------------------------------------------------------- sampler
def sample(
    score_fn: Callable,
    noise: Tensor,
    steps: int = 64,
    corrections: int = 0,
    tau: float = 1.0,
    z_draws: Optional[List[Tensor]] = None,
    eta: float = 1e-3,
) -> Tensor:
    """src/thor/pipelines.py:41-97.  Predictor: x <- mu(t-dt) x0_hat + sigma(t-dt) eps; corrector
    (Langevin, global-mean step): delta = tau / mean(eps^2); x <- x - (delta eps + sqrt(2 delta) z) sigma(t-dt).
    ``z_draws`` supplies the corrector normals in call order (steps*corrections tensors)."""
    x = noise.clone()
    ts = torch.linspace(1, 0, steps + 1).to(x.dtype)
    dt = 1 / steps
    zi = 0
    with torch.no_grad():
        for t in ts[:-1]:
            eps = score_fn(x, t)
            x0 = (x - sigma(t, eta) * eps) / mu(t, eta)
            x = mu(t - dt, eta) * x0 + sigma(t - dt, eta) * eps
            for _ in range(corrections):
                z = z_draws[zi]
                zi += 1
                eps = score_fn(x, t - dt)
                delta = tau / eps.square().mean()
                x = x - (delta * eps + torch.sqrt(2 * delta) * z) * sigma(t - dt, eta)
            if torch.isnan(x).any():
                raise ValueError("NaN detected in sample")
    return x.reshape(noise.shape)
