"""Time-sharded, device-resident sampling of ONE long trajectory across the GPUs of a node.

The reference keeps the whole ``(L, F, H, W)`` state on the host and moves every window batch over PCIe
(``src/thor/score.py:156-185``); ensemble members are its only parallel axis (``exp/downscaling.py:96-99``), so a
single 8737-frame member cannot use more than one GPU.  Here the TIME axis is sharded:

* rank r owns a contiguous run of frames ``[s_r, e_r)`` of the trajectory, resident in its HBM for the whole run;
* before every score evaluation each rank receives the ``k`` frames either side of its run from its two neighbours
  (point-to-point over xGMI: ``k*F*H*W`` fp32 per neighbour, 1.5 MB for the default k=6, F=4, 128x128) -- a window
  centred on an owned frame reaches ``k`` frames left and right and no further (``src/thor/score.py:68-74``);
* the windows centred on owned frames go through the network exactly as in ``BatchedScoreFunction``; ``fold``
  (``src/thor/score.py:76-88``) keeps window centres, plus the head of the first and the tail of the last window, which
  live on the first and last rank;
* the predictor is frame-local; the corrector's step size uses the GLOBAL mean of eps^2 (``src/thor/pipelines.py:84``):
  one scalar all-reduce; the Gaussian-likelihood guidance for ``A = AvgPool2d(s) o [::t]`` is frame-local
  (``exp/downscaling.py:129-132``) and only needs the global index of the first owned frame.

Communication uses ``torch.distributed`` (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import ops
from .pipelines import SDAPipeline
from .score_fn import BatchedScoreFunction, PoolStrideOperator, per_channel_std


def partition_frames(length: int, world: int, markov_order: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced frame ranges, one per rank.  Every rank must own at least ``markov_order`` frames so that a
    halo comes from the direct neighbour only."""
    base, rem = divmod(length, world)
    bounds, s = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        bounds.append((s, s + n))
        s += n
    if world > 1 and min(e - s for s, e in bounds) < max(markov_order, 1):
        raise ValueError(f"{length} frames over {world} ranks leaves a rank with fewer than k={markov_order} frames")
    if length < 2 * markov_order + 1:
        raise ValueError(f"trajectory of {length} frames is shorter than the window {2 * markov_order + 1}")
    return bounds


class TimeShardedScoreFunction(BatchedScoreFunction):
    """Score function over the frames ``[s, e)`` this rank owns of a length-``length`` trajectory.  ``__call__(x_own, t)``
    takes and returns the owned frames only; halos are exchanged inside."""

    def __init__(self, unet, markov_order: int, length: int, batch_size: int = 16, device=None, process_group=None,
                 rank: Optional[int] = None, world: Optional[int] = None, **kwargs):
        super().__init__(unet, markov_order, batch_size=batch_size, device=device, **kwargs)
        self.pg = process_group
        on = dist.is_available() and dist.is_initialized()
        self.rank = (dist.get_rank(process_group) if on else 0) if rank is None else rank
        self.world = (dist.get_world_size(process_group) if on else 1) if world is None else world
        self.length = int(length)
        self.bounds = partition_frames(self.length, self.world, markov_order)
        self.s, self.e = self.bounds[self.rank]
        self._guide = None

    # -------------------------------------------------------------------------------------------- halo exchange
    def _global(self, r: int) -> int:
        return r if self.pg is None else dist.get_global_rank(self.pg, r)

    def post_halo_exchange(self, x_own: torch.Tensor):
        """Start the exchange: -> (frames [s-kl, e+kr) as one tensor whose halo parts are still in flight, kl, pending work handles).
        kl / kr = k except at the ends of the trajectory.  The owned frames are in place; the halo frames are valid only after
        every handle has been waited for."""
        k = self.markov_order
        left = self.rank - 1 if self.rank > 0 else None
        right = self.rank + 1 if self.rank + 1 < self.world else None
        if (left is None and right is None) or k == 0:
            return x_own, 0, []
        n = x_own.shape[0]
        kl, kr = (k if left is not None else 0), (k if right is not None else 0)
        ext = torch.empty((kl + n + kr,) + tuple(x_own.shape[1:]), dtype=x_own.dtype, device=x_own.device)
        ext[kl:kl + n].copy_(x_own)
        p2p = []
        if left is not None:
            p2p.append(dist.P2POp(dist.isend, x_own[:k].contiguous(), self._global(left), group=self.pg))
            p2p.append(dist.P2POp(dist.irecv, ext[:k], self._global(left), group=self.pg))
        if right is not None:
            p2p.append(dist.P2POp(dist.isend, x_own[n - k:].contiguous(), self._global(right), group=self.pg))
            p2p.append(dist.P2POp(dist.irecv, ext[kl + n:], self._global(right), group=self.pg))
        return ext, kl, dist.batch_isend_irecv(p2p)

    def exchange_halos(self, x_own: torch.Tensor) -> Tuple[torch.Tensor, int]:
        """-> (frames [s-kl, e+kr) as one contiguous tensor, kl), halos landed."""
        ext, kl, works = self.post_halo_exchange(x_own)
        for w in works:
            w.wait()
        return ext, kl

    # -------------------------------------------------------------------------------------------- score
    overlap_halo = True  # evaluate the windows that lie inside the owned frames while the halo frames are in flight

    def __call__(self, x_own, t):
        if x_own.shape[0] != self.e - self.s:
            raise ValueError(f"rank {self.rank} owns frames [{self.s}, {self.e}) but got {x_own.shape[0]}")
        x_own = x_own.to(device=self.device, dtype=torch.float32).contiguous()
        if self._guide is not None and self._guide["exact"]:
            return self._call_exact(x_own, t)
        k, n = self.markov_order, x_own.shape[0]
        ext, kl, works = self.post_halo_exchange(x_own)
        nwin = ext.shape[0] - 2 * k
        # Window i reads ext[i : i + 2k + 1].  Windows [kl, kl + n - 2k) touch owned frames only: they run now; the kl windows in
        # front of them and the kr behind need halo frames and run once those have landed (the P2P transfers -- 1.5 MB per neighbour
        # at the default size -- overlap the interior windows' network evaluations instead of preceding all of them).
        lo, hi = kl, max(kl, kl + n - 2 * k)
        if works and self.overlap_halo and hi > lo and not self.use_graphs and (ext.is_cuda or ops.EMULATED):
            eps_ext = torch.empty_like(ext)
            self.score_fn(ext, t, ranges=[(lo, hi - lo)], out=eps_ext)
            for w in works:
                w.wait()
            self.score_fn(ext, t, ranges=[(0, lo), (hi, nwin - hi)], out=eps_ext)
        else:
            for w in works:
                w.wait()
            eps_ext = self.score_fn(ext, t)  # windows over the extended run; head/tail writes into halo frames are dropped below
        eps = eps_ext[kl:kl + x_own.shape[0]]
        if kl or eps_ext.shape[0] != x_own.shape[0]:
            eps = eps.contiguous()
        if self._guide is not None:
            self._apply_guidance(x_own, eps, t)
        return eps

    def _call_exact(self, x_own, t):
        """``exact_grad=True`` (the API default, src/thor/score.py:44): eps - sigma dlog p/dx with the network inside the derivative
        (src/thor/score.py:28-35,48-57).  log p is a sum over observed frames, each rank holds the terms of ITS observed frames; the
        estimate x0 of an owned frame reads the k frames either side of it, so a rank's terms also have a gradient with respect to its
        HALO frames -- which belong to the neighbours.  Forward halo exchange, local reverse pass over own + halo frames, then the
        reverse exchange: the gradient of my halo copies goes to their owners, theirs of my boundary frames comes back and is added."""
        g = self._guide
        k, n = self.markov_order, x_own.shape[0]
        ext, kl = self.exchange_halos(x_own)
        tt = torch.as_tensor(t).to(self.device)
        mu, sigma = self.noise_process.mu(tt), self.noise_process.sigma(tt)
        with torch.enable_grad():
            xg = ext.detach().requires_grad_(True)
            eps_ext = self.score_fn(xg, tt)  # differentiable route: unfold -> module (one autograd node per window batch) -> fold
            eps_own = eps_ext[kl:kl + n]
            if g["nobs"] > 0:
                x0 = (xg[kl:kl + n] - sigma * eps_own) / mu
                err = g["y"] - g["A"]._pool(x0[g["off"]:: g["A"].t_step][: g["nobs"]])
                F = x_own.shape[1]
                sd = g["std"].reshape(1, -1, 1, 1)
                gm = g["gamma"].reshape(1, F, 1, 1) if isinstance(g["gamma"], torch.Tensor) else g["gamma"]
                logp = -(err ** 2 / (sd ** 2 + gm * (sigma / mu) ** 2)).sum() / 2
                (J_ext,) = torch.autograd.grad(logp, xg)
            else:  # no observed frame here: this rank still takes part in the exchange below
                J_ext = torch.zeros_like(ext)
        J = J_ext[kl:kl + n].clone()
        left = self.rank - 1 if self.rank > 0 else None
        right = self.rank + 1 if self.rank + 1 < self.world else None
        p2p, rl, rr = [], None, None
        if left is not None:
            rl = torch.empty_like(J[:k])
            p2p += [dist.P2POp(dist.isend, J_ext[:kl].contiguous(), self._global(left), group=self.pg),
                    dist.P2POp(dist.irecv, rl, self._global(left), group=self.pg)]
        if right is not None:
            rr = torch.empty_like(J[n - k:])
            p2p += [dist.P2POp(dist.isend, J_ext[kl + n:].contiguous(), self._global(right), group=self.pg),
                    dist.P2POp(dist.irecv, rr, self._global(right), group=self.pg)]
        if p2p:
            for w in dist.batch_isend_irecv(p2p):
                w.wait()
        if rl is not None:
            J[:k] += rl
        if rr is not None:
            J[n - k:] += rr
        return (eps_own.detach() - sigma * J).contiguous()

    def condition_on(self, *, A, y, std, gamma=1e-2, exact_grad=False):
        """Same keywords as ``src/thor/score.py:44-60``; ``y`` is the GLOBAL observation ``A(x)`` of the whole trajectory.
        The frame-local operator of the reference's experiments shards over time; ``exact_grad=True`` adds a reverse halo exchange of
        the gradient (``_call_exact``).  An arbitrary callable ``A`` over the whole trajectory cannot be split and is refused.
        NOTE the default here is ``exact_grad=False`` (what every shipped experiment config sets), the reference API's is True."""
        if not isinstance(A, PoolStrideOperator):
            raise NotImplementedError("time-sharded guidance needs the frame-local operator A = PoolStrideOperator")
        if self._guide is not None:
            print("Warning: Overwriting old conditioning")
        t_step = A.t_step
        first = -(-self.s // t_step) * t_step  # first observed global frame >= s
        nobs = 0 if first >= self.e else (self.e - 1 - first) // t_step + 1
        i0 = first // t_step
        y_loc = y[i0:i0 + nobs].to(device=self.device, dtype=torch.float32).contiguous()
        std = per_channel_std(std, y)
        if std is None:
            raise NotImplementedError("time-sharded guidance takes a scalar std or one value per variable, shape (1, F, 1, 1)")
        std = std.to(self.device)
        gam = per_channel_std(gamma, y)  # a float, or one value per variable as exp/downscaling.py:228-233 builds it
        if gam is None:
            raise NotImplementedError("time-sharded guidance takes a scalar gamma or one value per variable, shape (1, F, 1, 1)")
        gam = float(gam) if gam.numel() == 1 else gam.to(self.device).contiguous()
        self._guide = dict(A=A, y=y_loc, std=std, gamma=gam, off=first - self.s, nobs=nobs, exact=bool(exact_grad))
        return self

    @property
    def is_conditioned(self):
        return self._guide is not None

    def _apply_guidance(self, x_own, eps, t):
        g = self._guide
        if g["nobs"] == 0:
            return
        _, F, H, W = x_own.shape
        std = g["std"].expand(F).contiguous() if g["std"].numel() == 1 else g["std"]
        mu, sigma = self.noise_process._mu_sigma_f(float(t))
        ops.guidance(x_own[g["off"]:], eps[g["off"]:], g["y"], std, g["nobs"], F, H, W, g["A"].s_step, g["A"].t_step, mu, sigma, g["gamma"])


def sample_time_sharded(pipeline: SDAPipeline, score_fn: TimeShardedScoreFunction, noise_own: torch.Tensor, steps: int = 64,
                        corrections: int = 0, tau: float = 1.0, z_draws: Optional[Sequence[torch.Tensor]] = None,
                        gather: bool = False) -> torch.Tensor:
    """``SDAPipeline.sample`` (``src/thor/pipelines.py:52-97``) on this rank's frames.  ``noise_own`` / the result are the
    owned frames ``[s, e)``; ``z_draws`` (tests) are corrector normals for the owned frames.  ``gather=True`` returns the
    whole trajectory on every rank (all-gather of unequal runs)."""
    dev, pg = score_fn.device, score_fn.pg
    x = noise_own.to(device=dev, dtype=torch.float32).clone().contiguous()
    n_own = x.numel()
    frame = n_own // max(x.shape[0], 1)
    n_global = frame * score_fn.length
    multi = score_fn.world > 1
    nan_flag = torch.zeros(1, dtype=torch.int32, device=dev)
    sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
    z = torch.empty_like(x) if corrections > 0 else None
    zs = iter(z_draws) if z_draws is not None else None
    ts = torch.linspace(1, 0, steps + 1).tolist()
    dt = 1 / steps
    with torch.no_grad():
        for i in range(steps):
            tf = ts[i]
            eps = score_fn(x, torch.tensor(tf, dtype=torch.float32))
            mu_t, sg_t = pipeline._mu_sigma_f(tf)
            mu_n, sg_n = pipeline._mu_sigma_f(tf - dt)
            ops.sampler_predict(x, eps, nan_flag, n_own, mu_n / mu_t, sg_n - mu_n * sg_t / mu_t)
            for _ in range(corrections):
                if zs is not None:
                    z.copy_(next(zs))
                else:
                    z.normal_()
                eps = score_fn(x, torch.tensor(tf - dt, dtype=torch.float32))
                sumsq.zero_()
                ops.sumsq(eps, sumsq, n_own)
                if multi:
                    dist.all_reduce(sumsq, op=dist.ReduceOp.SUM, group=pg)
                    sumsq.mul_(n_own / n_global)  # the kernel divides by its own element count: hand it the global mean
                ops.sampler_correct(x, eps, z, sumsq, nan_flag, n_own, tau, sg_n)
    if multi:
        dist.all_reduce(nan_flag, op=dist.ReduceOp.MAX, group=pg)
    if int(nan_flag.item()) != 0:
        raise ValueError("NaN detected in sample")
    if not gather:
        return x
    if not multi:
        return x
    parts = [torch.empty((e - s,) + tuple(x.shape[1:]), dtype=x.dtype, device=dev) for s, e in score_fn.bounds]
    # runs differ in length by at most one frame: pad to the longest for the collective
    longest = max(e - s for s, e in score_fn.bounds)
    pad = torch.zeros((longest,) + tuple(x.shape[1:]), dtype=x.dtype, device=dev)
    pad[: x.shape[0]].copy_(x)
    bufs = [torch.empty_like(pad) for _ in score_fn.bounds]
    dist.all_gather(bufs, pad, group=pg)
    for p, b in zip(parts, bufs):
        p.copy_(b[: p.shape[0]])
    return torch.cat(parts, 0)
