"""Drop-in for ``thor.pipelines.SDAPipeline`` (src/thor/pipelines.py:8-97): VP-cosine noise process, epsilon-MSE
loss and the predictor(-corrector) sampler.

``pipeline_kwargs.class_name`` (train.py:184) can point here.  Same methods and signatures.  The sampler keeps
the trajectory on the GPU and uses the fused HIP update kernels (csrc/sampler.hip) whenever the score function is
one of ``climate2weather_amd.score_fn``'s; with any other callable it runs the same update rule with torch ops on
whatever device the state lives on.
"""
from __future__ import annotations

import math
import time
from typing import Optional

import torch

from . import ops


def _engine_module(net):
    """The engine-backed ScoreUNet behind ``net`` -- the module itself or what a DDP / Fabric wrapper holds (``.module``) -- or None."""
    m, hops = net, 0
    while m is not None and not hasattr(m, "_get_engine") and hops < 4:
        m, hops = getattr(m, "module", None), hops + 1
    return m if m is not None and hasattr(m, "_get_engine") else None


class SDAPipeline:
    # How ``loss`` runs when ``net`` is (a wrapper around) this package's ScoreUNet on a GPU -- a CLASS attribute, so that the
    # instance's __dict__ stays {"eta"}: the reference checkpoints ``pipeline.__dict__`` (src/thor/checkpoint.py:13-35).
    #   True   x_t = mu x + sigma eps, the network and (eps_pred - eps)^2 as ONE autograd node (score._forward_loss): eps is a
    #          counter-based stream regenerated inside the input-conversion and loss kernels from a per-call seed, so neither eps nor
    #          x_t nor eps_pred ever exists as a (B,C,H,W) tensor; the returned loss tensor answers .mean() from a sum its kernel made
    #   "eps"  the same node with eps = torch.randn_like(x) drawn as the reference draws it (reproduces torch's stream; CPU tests)
    #   False  the reference's own tensor arithmetic around net(x_t, t)
    fused_loss = True

    def __init__(self, eta: float = 1e-3):
        self.eta = eta  # numerical-stability floor of the schedule (src/thor/pipelines.py:9-11)

    # ---- schedule (src/thor/pipelines.py:13-20)
    def alpha(self, t):
        return torch.cos(math.acos(math.sqrt(self.eta)) * t) ** 2

    def mu(self, t):
        return self.alpha(t)

    def sigma(self, t):
        return (1 - self.alpha(t) ** 2 + self.eta**2).sqrt()

    def _mu_sigma_f(self, t: float):
        a = math.cos(math.acos(math.sqrt(self.eta)) * t) ** 2
        return a, math.sqrt(1 - a * a + self.eta**2)

    # ---- training side (src/thor/pipelines.py:22-35)
    def forward(self, x, t):
        eps = torch.randn_like(x)
        return self.mu(t) * x + self.sigma(t) * eps, eps

    def loss(self, net, x, forcing=None):
        core = _engine_module(net) if self.fused_loss and forcing is None else None
        if core is not None and len(x.shape) == 4 and not getattr(x, "requires_grad", False) and (x.is_cuda or self.fused_loss == "eps"):
            # the call still goes through ``net`` (a DDP wrapper prepares its gradient hooks in forward: training_loop.py:116,375-377)
            t = torch.rand(x.shape[0], 1, 1, 1, dtype=torch.float32, device=x.device)
            if self.fused_loss == "eps":
                eps = torch.randn(tuple(x.shape), dtype=torch.float32, device=x.device)
            else:  # one 62-bit seed from torch's CPU generator: reproducible under torch.manual_seed, no device synchronisation
                eps = int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64).item())
            core.__dict__["_loss_request"] = dict(eps=eps, eta=self.eta)
            try:
                return net(x, t)
            finally:
                core.__dict__.pop("_loss_request", None)
        if hasattr(x, "materialize"):  # data.WindowBatch
            x = x.materialize()
        t = torch.rand(x.shape[0], 1, 1, 1, dtype=x.dtype, device=x.device)
        xt, eps = self.forward(x, t)
        eps_pred = net(xt, t, forcing=forcing)
        return (eps_pred - eps) ** 2

    def pred_eps(self, score_fn, x, t):
        return score_fn(x, t)

    # ---- sampling side (src/thor/pipelines.py:41-97)
    def _sample_step(self, score_fn, x, t, dt, proc_x0=None):
        eps_pred = self.pred_eps(score_fn, x, t)
        pred_x0 = (x - self.sigma(t) * eps_pred) / self.mu(t)
        if proc_x0 is not None:
            pred_x0 = proc_x0(pred_x0)
        return self.mu(t - dt) * pred_x0 + self.sigma(t - dt) * eps_pred

    def sample(self, score_fn, noise, steps: int = 64, corrections: int = 0, tau: float = 1.0, proc_x0=None, device=None,
               show_progressbar: bool = True, z_draws=None):
        """Same contract as the reference.  ``device=None`` follows the reference's default (CPU-resident state) unless the
        score function is device-resident, in which case the state stays on its GPU.  ``z_draws`` (extension, tests):
        iterable of corrector normals to use instead of drawing them.  ``noise`` of shape (M, L, F, H, W) (extension) co-samples
        M ensemble members: every member evolves exactly as it would alone (the corrector's step size uses the member's own
        mean(eps^2)), but their windows share the network batches."""
        fused = getattr(score_fn, "device_resident", False) and proc_x0 is None
        if device is None:
            device = score_fn.device if fused else torch.device("cpu")
        device = torch.device(device)
        fused = fused and device.type == "cuda"
        shape = noise.shape
        x = noise.to(device=device).clone() if fused else noise.to(device=device)
        time_steps = torch.linspace(1, 0, steps + 1).to(dtype=x.dtype, device=device)
        dt = 1 / steps
        start = time.time()
        zs = iter(z_draws) if z_draws is not None else None
        z = torch.empty_like(x) if corrections > 0 else None
        nan_flag = torch.zeros(1, dtype=torch.int32, device=device) if fused else None
        # The reference raises in the step that produced the NaN (src/thor/pipelines.py:90-91: an isnan over the state = a device
        # synchronisation per step).  Here the update kernels or a flag on the device; a one-thread kernel publishes the flag into pinned
        # host memory behind every step (ops.HostRing: the mechanism loss.item() uses) and the host reads step i - 1's publication after
        # it has ENQUEUED step i: it never drains a stream, runs at most one step ahead of the device, and raises one step late.
        ring = ops.HostRing() if fused else None
        pending = None
        sumsq = torch.zeros(1, dtype=torch.float32, device=device) if fused and corrections > 0 else None
        ts_host = torch.linspace(1, 0, steps + 1).tolist()
        iterator = range(steps)
        if show_progressbar:
            try:
                from tqdm.auto import tqdm
                iterator = tqdm(iterator, desc="Sampling")
            except Exception:  # pragma: no cover
                pass
        with torch.no_grad():
            for i in iterator:
                t = time_steps[i]
                if fused:
                    tf = ts_host[i]
                    t = torch.tensor(tf, dtype=torch.float32)  # host scalar: no device->host sync per step
                    x = x.float().contiguous()
                    eps = score_fn(x, t)
                    mu_t, sg_t = self._mu_sigma_f(tf)
                    mu_n, sg_n = self._mu_sigma_f(tf - dt)
                    n = x.numel()
                    ops.sampler_predict(x, eps, nan_flag, n, mu_n / mu_t, sg_n - mu_n * sg_t / mu_t)
                    for _ in range(corrections):
                        if zs is not None:
                            z.copy_(next(zs))
                        else:
                            z.normal_()
                        eps = score_fn(x, t - dt)
                        if x.dim() == 5:  # co-sampled members: delta = tau / mean(eps^2) per member (src/thor/pipelines.py:84 on each)
                            nm = n // x.shape[0]
                            for m in range(x.shape[0]):
                                sumsq.zero_()
                                ops.sumsq(eps[m], sumsq, nm)
                                ops.sampler_correct(x[m], eps[m], z[m], sumsq, nan_flag, nm, tau, sg_n)
                        else:
                            sumsq.zero_()
                            ops.sumsq(eps, sumsq, n)
                            ops.sampler_correct(x, eps, z, sumsq, nan_flag, n, tau, sg_n)
                    if pending is not None:  # the flag as it stood behind the PREVIOUS step
                        bits = ring.read_bits(*pending)
                        if bits is None:  # nothing arrived in time (or the slot was reused): say so instead of reading it as "no NaN"
                            import warnings
                            warnings.warn("SDAPipeline.sample: the NaN flag of the previous step did not arrive in pinned memory in time; "
                                          "that step's check is skipped (the final check behind the loop still runs)", RuntimeWarning, stacklevel=2)
                        elif bits:
                            raise ValueError("NaN detected in sample")
                    pending = ring.publish(nan_flag)
                else:
                    x = self._sample_step(score_fn, x, t, dt, proc_x0=proc_x0)
                    for _ in range(corrections):
                        if zs is not None:
                            z.copy_(next(zs))
                        else:
                            z.normal_()
                        eps = score_fn(x, t - dt)
                        red = tuple(range(-len(shape), 0)) if len(shape) != 5 else (-4, -3, -2, -1)  # co-sampled members: per member
                        delta = tau / eps.square().mean(dim=red, keepdim=True)
                        x = x - (delta * eps + torch.sqrt(2 * delta) * z) * self.sigma(t - dt)
                        del eps
                    if torch.isnan(x).any():
                        raise ValueError("NaN detected in sample")
        if fused and int(nan_flag.item()) != 0:  # the last step's flag: the trajectory's end is a synchronisation anyway
            raise ValueError("NaN detected in sample")
        total = time.time() - start
        print(f"Total sampling time: {total:.2f} s  = {total / 60:.3f} min = {total / 3600:.4f} h")
        return x.reshape(shape)
