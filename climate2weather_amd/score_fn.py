"""Drop-ins for ``thor.score`` (src/thor/score.py:7-185): sliding-window score functions with optional
Gaussian-likelihood guidance.

Same constructors and methods as the reference (``DefaultScoreFunction(unet, markov_order, noise_process=...)``,
``BatchedScoreFunction(unet, markov_order, batch_size, device, noise_process=...)``, ``condition_on(A=, y=, std=,
gamma=, exact_grad=)``).  Differences are all below the API: the trajectory stays in HBM, windows are gathered
straight into the network's NHWC input by a HIP kernel (no ``unfold`` materialisation, no PCIe round trip per batch,
src/thor/score.py:165-183), only the kept frames are written back, and for the reference's own measurement operator
(``PoolStrideOperator`` = AvgPool2d(s) o x[::t], exp/downscaling.py:129-132) with ``exact_grad=False`` the guidance
term is one fused kernel instead of ``torch.func.jacrev``.
"""
from __future__ import annotations

import contextlib
import os
from typing import Optional

import torch

from . import ops
from .ops import TORCH_DTYPE


class PoolStrideOperator:
    """The reference's measurement operator ``A(x) = AvgPool2d(s_step)(x[::t_step])`` (exp/downscaling.py:129-132) as an
    object the score functions can recognise; calling it applies the same torch ops."""

    def __init__(self, s_step: int, t_step: int):
        self.s_step, self.t_step = int(s_step), int(t_step)
        self._pool = torch.nn.AvgPool2d(self.s_step)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and not (torch.is_grad_enabled() and x.requires_grad) \
                and x.shape[-1] % self.s_step == 0 and x.shape[-2] % self.s_step == 0 and not _wrapped(x):
            L, F, H, W = x.shape
            nobs = (L + self.t_step - 1) // self.t_step
            y = torch.empty((nobs, F, H // self.s_step, W // self.s_step), dtype=torch.float32, device=x.device)
            ops.pool_stride(x.contiguous(), y, nobs, F, H, W, self.s_step, self.t_step)
            return y
        return self._pool(x[:: self.t_step])  # differentiable / host path: the same torch ops as the reference


def per_channel_std(std, y) -> Optional[torch.Tensor]:
    """``std`` (likewise ``gamma``) as the fused guidance kernel indexes it -- one value, or one per variable -- or None when it
    broadcasts against ``err = y - A(x0)`` of shape (nobs, F, h, w) in any other way (src/thor/score.py:55: ``std**2`` and ``gamma``
    broadcast like any tensor; the experiments pass floats or (1, F, 1, 1) tensors, exp/downscaling.py:219-242)."""
    std = torch.as_tensor(std, dtype=torch.float32)
    if std.numel() == 1:
        return std.reshape(1)
    F = int(y.shape[-3]) if hasattr(y, "shape") and len(y.shape) >= 3 else None
    if F is not None and std.numel() == F and std.dim() >= 3 and tuple(std.shape[-3:]) == (F, 1, 1):
        return std.reshape(F)
    return None


per_channel = per_channel_std  # the same rule serves gamma


class AbstractScoreFunction:
    device_resident = True

    def __init__(self, unet, noise_process, unet_kwargs=None):
        self.unet = unet
        self.noise_process = noise_process
        self.unet_kwargs = unet_kwargs if unet_kwargs is not None else {}
        self.likelihood = None
        self._fused_guidance = None
        self.unet.eval()

    @property
    def is_conditioned(self):
        return self.likelihood is not None

    def net_forward(self, x, t):
        return self.unet(x, t)

    def __call__(self, x, t):
        if not self.is_conditioned:
            return self.score_fn(x, t)
        if self._fused_guidance is not None:
            return self._guided_fused(x, t)
        if x.dim() == 5:  # co-sampled members: log p sums over members, its gradient is per member -- member by member
            return torch.stack([self(xm, t) for xm in x], 0)
        # eps - sigma * d(log p)/dx  (src/thor/score.py:24-35).  log p is a scalar, so its Jacobian is one reverse pass:
        # plain autograd gives what the reference's jacrev(chunk_size=1) gives.
        with torch.enable_grad():
            xg = x.detach().requires_grad_(True)
            logp, (epsilon, sigma_t) = self.likelihood(xg, t)
            (J,) = torch.autograd.grad(logp, xg)
        return epsilon.detach() - sigma_t * J

    def score_fn(self, x, t):
        raise NotImplementedError

    def condition_on(self, *, A, y, std, gamma=1e-2, exact_grad=True):
        if self.likelihood is not None:
            print("Warning: Overwriting old conditioning")

        def log_p(x, t):  # src/thor/score.py:48-57
            mu, sigma = self.noise_process.mu(t), self.noise_process.sigma(t)
            with torch.set_grad_enabled(exact_grad):
                eps_pred = self.noise_process.pred_eps(self.score_fn, x, t)
            x0_pred = (x - sigma * eps_pred) / mu
            # the observation follows the state: the reference keeps both on the host, the device-resident sampler in HBM
            yy = y.to(x.device) if isinstance(y, torch.Tensor) else y
            sd = std.to(x.device) if isinstance(std, torch.Tensor) else std
            gm = gamma.to(x.device) if isinstance(gamma, torch.Tensor) else gamma  # (1, F, 1, 1) for list-valued settings (exp/downscaling.py:228-233)
            err = yy - A(x0_pred)
            var = sd**2 + gm * (sigma / mu) ** 2
            return -(err**2 / var).sum() / 2, (eps_pred, sigma)

        self.likelihood = log_p
        self._fused_guidance = None
        if isinstance(A, PoolStrideOperator) and not exact_grad:
            std_c, gam_c = per_channel_std(std, y), per_channel(gamma, y)
            if std_c is not None and gam_c is not None:  # any other broadcastable std / gamma (per pixel, per observation, ...) takes the autograd path above
                self._fused_guidance = dict(A=A, y=y, std=std_c, gamma=float(gam_c) if gam_c.numel() == 1 else gam_c)
        return self

    def _guided_fused(self, x, t):
        g = self._fused_guidance
        dev = getattr(self, "device", x.device)
        xd = x.to(device=dev, dtype=torch.float32).contiguous()
        L, F, H, W = xd.shape[-4:]
        if g.get("dev") != dev:
            g["y_dev"] = g["y"].to(device=dev, dtype=torch.float32).contiguous()
            std = g["std"].to(dev)
            g["std_dev"] = (std.expand(F) if std.numel() == 1 else std).contiguous()
            g["gamma_dev"] = g["gamma"].to(dev).contiguous() if isinstance(g["gamma"], torch.Tensor) else g["gamma"]  # one per variable, or a float
            g["dev"] = dev
        eps = self.score_fn(xd, t)
        mu, sigma = self.noise_process._mu_sigma_f(float(t))
        nobs = g["y_dev"].shape[0]
        if xd.dim() == 5:  # co-sampled ensemble members: the same observation guides every member, frame-locally
            for m in range(xd.shape[0]):
                ops.guidance(xd[m], eps[m], g["y_dev"], g["std_dev"], nobs, F, H, W, g["A"].s_step, g["A"].t_step, mu, sigma, g["gamma_dev"])
        else:
            ops.guidance(xd, eps, g["y_dev"], g["std_dev"], nobs, F, H, W, g["A"].s_step, g["A"].t_step, mu, sigma, g["gamma_dev"])
        return eps if x.device == dev else eps.to(x.device)


class _WindowScore(AbstractScoreFunction):
    def __init__(self, unet, markov_order, batch_size=None, device=None, **kwargs):
        super().__init__(unet=unet, **kwargs)
        self.markov_order = markov_order
        self.batch_size = batch_size
        if device is None:
            try:
                device = next(unet.parameters()).device
            except StopIteration:  # pragma: no cover
                device = torch.device("cuda")
        self.device = torch.device(device)
        # window batches of one score evaluation alternate between this many HIP streams (1: the caller's stream only);
        # C2W_SCORE_STREAMS (read once, here) or the attribute
        self.num_streams = int(os.environ.get("C2W_SCORE_STREAMS", type(self).num_streams))
        if "C2W_WINDOW_BATCH_FLOOR" in os.environ:
            self.window_batch_floor = int(os.environ["C2W_WINDOW_BATCH_FLOOR"])

    # reference-compatible helpers (src/thor/score.py:68-88)
    def unfold(self, x):
        w = 2 * self.markov_order + 1
        return x.unfold(0, w, 1).movedim(-1, 1).flatten(1, 2)

    def fold(self, x):
        k = self.markov_order
        x = x.unflatten(1, (2 * k + 1, -1))
        return torch.cat((x[0, :k], x[:, k], x[-1, -k:]), dim=0)

    def score_fn(self, x, t, ranges=None, out=None):
        """eps(x, t) over the whole trajectory x: (L, F, H, W) -- or over M co-sampled ensemble members (M, L, F, H, W):
        the windows of all members form one list that is fed to the network ``batch_size`` at a time, so short
        trajectories (37 windows at L = 49) still fill the chip (an extension: the reference samples members one by one,
        exp/downscaling.py:248-265).
        ``ranges`` (one trajectory only): [(first window, count)] -- evaluate only these windows, writing their kept frames into
        ``out`` (an eps buffer of x's shape from an earlier call); the time-sharded sampler runs the windows that need no halo
        while the halo frames are still in flight (sharded.py)."""
        if ranges is not None and (x.dim() != 4 or self.use_graphs):
            raise ValueError("window ranges apply to one eagerly evaluated trajectory")
        if not _engine_ready(self.unet) or torch.is_grad_enabled() and x.requires_grad or _wrapped(x):
            if ranges is not None:
                raise ValueError("window ranges need the engine path")
            if x.dim() == 5:
                return torch.stack([self._score_generic(xm, t) for xm in x], 0)
            return self._score_generic(x, t)  # differentiable / foreign-network path: same math through the module call
        k = self.markov_order
        w = 2 * k + 1
        src_dev = x.device
        xd = x.to(device=self.device, dtype=torch.float32).contiguous()
        M = xd.shape[0] if xd.dim() == 5 else 1
        L, F, H, W = xd.shape[-4:]
        nwin = L - w + 1
        if nwin < 1:
            raise ValueError(f"trajectory of {L} frames is shorter than the window {w}")
        eng = self.unet._get_engine()
        dt = self.unet.compute_dtype()
        lay = eng.layout
        if lay.in_channels != w * F:
            raise ValueError(f"network expects {lay.in_channels} channels, window gives {w * F}")
        total = M * nwin
        bs = min(max(self.batch_size or total, self._window_floor(H * W)), total)
        if xd.dim() == 5 or int(self.window_batch_floor) > 0:
            # equal batches instead of full ones and a ragged tail (co-sampled members, 148 windows: 2 x 74, not 128 + 20; L = 1037
            # under the floor: 5 x 205, not 4 x 256 + 1 -- a one-window network call costs 1.5 ms of launches for nothing)
            bs = -(-total // -(-total // bs))
        if self.use_graphs and xd.is_cuda and src_dev == self.device and xd.dim() == 4:
            return self._score_graphed(xd, t, eng, dt, lay, k, w, nwin, bs)
        eps = torch.empty_like(xd) if out is None else out
        xs, es = xd.view(M, L, F, H, W), eps.view(M, L, F, H, W)
        td = torch.as_tensor(t).to(self.device)
        HW = H * W
        # Window batches are independent: with more than one they alternate between HIP streams, so that one batch's low-resolution
        # levels (8x8: one workgroup per CU), HBM-bound passes and last rounds of workgroups overlap the other's full-chip
        # convolutions.  Everything the batches share is read-only and prepared on the caller's stream first.
        if ranges is None:
            batches = [(g0, min(bs, total - g0)) for g0 in range(0, total, bs)]
        else:
            batches = [(g0, min(bs, a + cnt - g0)) for a, cnt in ranges if cnt > 0 for g0 in range(a, a + cnt, bs)]
            if any(a < 0 or a + cnt > nwin for a, cnt in ranges):
                raise ValueError(f"window range outside [0, {nwin})")
        streams = self._side_streams(len(batches)) if xd.is_cuda else []
        if streams:
            eng.prepare_forward(dt)
            main = torch.cuda.current_stream()
            for st in streams:
                st.wait_stream(main)
        for bi, (g0, ng) in enumerate(batches):
            with (torch.cuda.stream(streams[bi % len(streams)]) if streams else contextlib.nullcontext()):
                xin = torch.empty((ng * HW, lay.cin_pad), dtype=TORCH_DTYPE[dt], device=self.device)
                segs, pos = [], 0
                while pos < ng:  # the batch's windows, member by member: (member, first window, count, row offset in the batch)
                    m, i0 = divmod(g0 + pos, nwin)
                    nw = min(nwin - i0, ng - pos)
                    ops.window_gather(xs[m], xin[pos * HW:], nw, F, HW, k, i0, lay.cin_pad, dt)
                    segs.append((m, i0, nw, pos))
                    pos += nw
                y = eng.forward(None, td, dt, x_nhwc=xin, shape=(ng, w * F, H, W), nhwc_out=True,
                                fold=dict(segs=[(es[m], i0, nw, pos) for m, i0, nw, pos in segs], k=k, F=F, nwin=nwin))
                if y is not None:  # the engine could not fold inside its output convolution: all w * F channels came back
                    for m, i0, nw, pos in segs:
                        ops.window_scatter(y[pos * HW:], es[m], nw, F, HW, k, i0, nwin, lay.cout_pad, dt)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        return eps if src_dev == self.device else eps.to(src_dev)

    # hipGraph replay of the launch sequence (BASELINE.json configs[4]: "hipGraph-captured sampler step").  A score evaluation
    # is ~200 short launches per window batch; at the shipped trajectory lengths (37 / 109 windows) the host-side launch path,
    # not the GPU, sets the step time.  Everything the kernels read is at fixed addresses: a persistent copy of the trajectory
    # (refreshed by one device copy per evaluation, so the captures outlive a trajectory: the members of an ensemble share them), a
    # persistent eps buffer and a one-element time buffer that is overwritten before each replay.
    use_graphs = False
    num_streams = 4  # window batches of one score evaluation alternate between this many HIP streams (1: the caller's stream only)
    # ``batch_size`` bounds activation memory in the reference (src/thor/score.py:156-185; the shipped configs say 32 and 128 for
    # 40-80 GB devices).  On the engine path it is a LOWER bound: a launch sequence carries at least this many windows of 128x128
    # pixels (fewer, in proportion, for larger fields), because the 16x16 and 8x8 levels of a 32-window batch leave three quarters
    # of the CUs without a workgroup (L = 1037, one member: 7.8 k window-forwards/s at 32 windows per batch, 9.1 k at 128, 9.3 k at
    # 256; profiles/r04_experiments.md section 15).  256 windows of bf16 activations are ~5 GB per stream.  The windows are
    # independent, so the result is the same trajectory; 0 (attribute, or C2W_WINDOW_BATCH_FLOOR read at construction) restores
    # exactly ``batch_size`` windows per network call.
    window_batch_floor = 256

    # bytes of activations one window of 128 x 128 pixels keeps alive during an inference forward of the default network, per byte of
    # element size (256 windows of bf16 ~ 5 GB: measured); the floor never raises a batch beyond what fits next to what is allocated
    _ACT_BYTES_PER_WINDOW_PER_ESZ = 10 << 20

    def _window_floor(self, pixels: int) -> int:
        n = int(self.window_batch_floor)
        if n <= 0:
            return 1
        floor = max(1, n * 128 * 128 // max(pixels, 1))
        configured = int(self.batch_size or 0)
        if configured and floor > configured and self.device.type == "cuda":
            # ``batch_size`` is the reference's bound on activation memory (src/thor/score.py:156-185): the floor overrides it only as far
            # as the device's FREE memory allows (a user who lowered batch_size to fit next to other allocations keeps fitting), and
            # says so once
            try:
                free, _ = torch.cuda.mem_get_info(self.device)
            except RuntimeError:
                free = None
            if free is not None:
                streams = max(1, int(self.num_streams))
                per_window = self._ACT_BYTES_PER_WINDOW_PER_ESZ * 2 * max(pixels, 1) // (128 * 128)
                fit = int(0.5 * free) // max(1, per_window * streams)
                floor = max(configured, min(floor, fit))
            if floor > configured and not self.__dict__.get("_floor_notice"):
                self.__dict__["_floor_notice"] = True
                print(f"BatchedScoreFunction: network calls carry {floor} windows (window_batch_floor; batch_size = {configured} is a lower "
                      f"bound on this device, C2W_WINDOW_BATCH_FLOOR=0 restores the reference's meaning)", flush=True)
        return floor

    def _side_streams(self, nbatches: int):
        n = min(int(self.num_streams), nbatches)
        if n < 2:
            return []
        pool = self.__dict__.setdefault("_streams", [])
        while len(pool) < n:
            pool.append(torch.cuda.Stream(device=self.device))
        return pool[:n]

    def _score_graphed(self, xd, t, eng, dt, lay, k, w, nwin, bs):
        L, F, H, W = xd.shape
        # replay never re-enters eng.forward(): re-read the Parameters' version counters here, so that weights written through the
        # Parameter objects since the capture (load_state_dict, torch.optim steps, EMA copies) invalidate the captured graphs and the
        # 16-bit shadow they read is rebuilt by the eager warm-up below
        eng.refresh_version()
        key = (L, F, H, W, bs, dt, eng._version())
        st = self._graphs.get(key) if hasattr(self, "_graphs") else None
        if st is None:
            if not hasattr(self, "_graphs"):
                self._graphs = {}
            self._graphs.clear()  # one shape at a time: new weights / another trajectory length invalidate the old captures
            xbuf = torch.empty_like(xd)  # the captured launches read the trajectory from here: every member / call replays the same graphs
            xbuf.copy_(xd)
            eps = torch.empty_like(xd)
            td = torch.zeros(1, dtype=torch.float32, device=self.device)
            td.fill_(float(t))

            def run(i0, nw):
                xin = torch.empty((nw * H * W, lay.cin_pad), dtype=TORCH_DTYPE[dt], device=self.device)
                ops.window_gather(xbuf, xin, nw, F, H * W, k, i0, lay.cin_pad, dt)
                y = eng.forward(None, td, dt, x_nhwc=xin, shape=(nw, w * F, H, W), nhwc_out=True,
                                fold=dict(segs=[(eps, i0, nw, 0)], k=k, F=F, nwin=nwin))
                if y is not None:
                    ops.window_scatter(y, eps, nw, F, H * W, k, i0, nwin, lay.cout_pad, dt)

            batches = [(i0, min(bs, nwin - i0)) for i0 in range(0, nwin, bs)]
            for i0, nw in batches[:1] + batches[-1:]:  # eager once per distinct batch size: lazy kernel attributes / weight casts
                run(i0, nw)
            torch.cuda.synchronize()
            graphs = []
            for i0, nw in batches:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    run(i0, nw)
                graphs.append(g)
            st = self._graphs[key] = dict(eps=eps, td=td, graphs=graphs, xbuf=xbuf)
        st["xbuf"].copy_(xd)  # 61 MB at the deep variant's size: 25 us against an 18.7 ms evaluation
        st["td"].fill_(float(t))
        for g in st["graphs"]:
            g.replay()
        return st["eps"]

    def _score_generic(self, x, t):
        win = self.unfold(x)
        bs = self.batch_size or win.shape[0]
        outs = [self.net_forward(chunk.to(self.device), torch.as_tensor(t).to(self.device)).to(x.device) for chunk in win.split(bs, 0)]
        return self.fold(torch.cat(outs, 0))


class DefaultScoreFunction(_WindowScore):
    """src/thor/score.py:63-93: every window in one batch."""

    def __init__(self, unet, markov_order, **kwargs):
        super().__init__(unet, markov_order, batch_size=None, **kwargs)


class BatchedScoreFunction(_WindowScore):
    """src/thor/score.py:96-185: windows fed ``batch_size`` at a time (bounds activation memory)."""

    def __init__(self, unet, markov_order, batch_size=16, device=None, **kwargs):
        super().__init__(unet, markov_order, batch_size=batch_size, device=device, **kwargs)
        print(f">>> Initialized batched score function to use device: {self.device}")


def _engine_ready(unet) -> bool:
    return hasattr(unet, "_get_engine")


def _wrapped(x) -> bool:
    try:
        return torch._C._functorch.is_functorch_wrapped_tensor(x)
    except Exception:  # pragma: no cover
        return False
