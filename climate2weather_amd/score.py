"""Drop-in for the reference's ``model.score.ScoreUNet`` (model/score.py:37-70).

Point ``network_kwargs.class_name`` (train.py:164-173, util.py:117-127) at
``climate2weather_amd.score.ScoreUNet`` and the reference's drivers get the MI355X engine: same constructor
keywords, same 228 state_dict keys and creation-order initialisation, same ``forward(x, t, forcing=None)``.

Precision follows the caller the way the reference's modules do: fp32 arithmetic unless ``torch.autocast`` is
active (the reference hard-codes Fabric ``precision="16-mixed"`` = fp16 autocast, train.py:98), in which case the
16-bit MFMA path of the autocast dtype runs (``torch.autocast("cuda", dtype=torch.float16)`` -> fp16, the reference's
own arithmetic type; the default / bfloat16 -> bf16).  ``net.precision = "fp32" | "bf16" | "fp16"`` pins it.
"""
from __future__ import annotations

import os

from typing import Optional

import torch

from .engine import Engine, Tape
from .nn import UNet
from .ops import DTYPE_BF16, DTYPE_F16, DTYPE_F32, TORCH_DTYPE


def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """model/score.py:14-34 on the GPU (HIP kernel); ``timesteps`` 1-D."""
    from . import ops
    t = timesteps.reshape(-1).float().contiguous()
    out = torch.empty((t.numel(), dim), dtype=torch.float32, device=t.device)
    ops.timestep_embedding(t, out, t.numel(), dim, max_period)
    return out.to(timesteps.dtype)


class _TapeHolder:
    """Carries the backward tape from ``forward`` to ``setup_context``.  It travels as a (non-tensor) INPUT of the autograd
    Function: ``setup_context`` receives the same input tuple however many times a functorch transform calls it, and the tape
    lives exactly as long as something references the holder or the context -- no global table, no finalizers (a context of
    a grad-disabled level can die between two ``setup_context`` calls of the same forward: seen under the reference's
    ``torch.func.jacrev(log_p)`` with ``exact_grad=False``)."""
    __slots__ = ("tape", "loss", "loss_sum", "forcing")

    def __init__(self, loss=None, forcing=None):
        self.tape = None
        self.loss = loss  # fused-loss request of SDAPipeline.loss: dict(eps=<int seed | (B,C,H,W) tensor>, eta=float), else None
        self.loss_sum = None
        self.forcing = forcing  # (B or 1, forcing_dim) conditioning vector of model/score.py:65-66 (a constant of the node), or None


class _ScoreUNetFn(torch.autograd.Function):
    """One autograd node for the whole network: forward and backward are the engine's hand-written HIP sequences.
    New-style (setup_context) so ``torch.func.jacrev`` / ``vjp`` can drive it (src/thor/score.py:28-33)."""

    @staticmethod
    def forward(x, t, net, dt, holder, *params):
        eng = net._get_engine()
        tape = Tape()
        if holder.loss is not None:  # noise process + network + unreduced loss in one node (src/thor/pipelines.py:27-35)
            y, holder.loss_sum = _forward_loss(eng, x, t, dt, holder.loss, tape)
        else:
            y = eng.forward(x, t, dt, tape=tape, want_dx=True, forcing=holder.forcing)
        holder.tape = tape
        return y

    @staticmethod
    def setup_context(ctx, inputs, output):
        x, t, net, dt, holder = inputs[:5]
        ctx.net = net
        ctx.tape = holder.tape
        ctx.x_needs_grad = bool(getattr(x, "requires_grad", False)) and holder.loss is None
        ctx.set_materialize_grads(False)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        from . import ops
        net = ctx.net
        eng: Engine = net._get_engine()
        lay = eng.layout
        tape = ctx.tape
        m = tape.meta
        n_params = len(lay.views)
        if gy is None:
            return (None, None, None, None, None) + (None,) * n_params
        # gradients of this call only: use a private flat buffer so autograd's accumulation semantics hold
        want_dw = any(ctx.needs_input_grad[5:])  # a frozen network (requires_grad_(False): the sampler's copy) needs no weight gradient
        saved = eng.flat_grad
        eng.flat_grad = torch.zeros_like(eng.flat) if want_dw else None
        try:
            dt = m["dt"]
            g_nhwc = torch.empty((m["B"] * m["H"] * m["W"], lay.cout_pad), dtype=TORCH_DTYPE[dt], device=gy.device)
            if "loss" in m:
                _loss_output_gradient(m, gy, g_nhwc, lay, dt)
            else:
                gy = gy.contiguous().float()
                ops.nchw_to_nhwc(gy, None, None, g_nhwc, m["B"], m["C"], m["H"] * m["W"], lay.cout_pad, dt)
            dx = eng.backward(tape, g_nhwc, want_dx=ctx.x_needs_grad, want_dw=want_dw)
            fg = eng.flat_grad
        finally:
            eng.flat_grad = saved
        if not want_dw:
            return (dx, None, None, None, None) + (None,) * n_params
        grads = tuple(torch.as_strided(fg, shape, strides, off) for (off, shape, strides) in lay.views.values())
        return (dx, None, None, None, None) + grads


class _SegCtrl:
    """What the nodes of one segmented call share (see _GradSegment)."""
    __slots__ = ("net", "holder", "dt", "nseg", "bounds", "owned", "y", "it", "fg", "events", "x_needs_grad")

    def __init__(self, net, holder, dt, nseg, bounds, owned):
        self.net, self.holder, self.dt, self.nseg, self.bounds, self.owned = net, holder, dt, nseg, bounds, owned
        self.y = self.it = self.fg = None
        self.events = {}
        self.x_needs_grad = False


class _GradSegment(torch.autograd.Function):
    """The same network call as _ScoreUNetFn, as a CHAIN of autograd nodes that deliver the parameter gradients in segments while
    the backward pass is still running -- for torch's DistributedDataParallel (fabric.setup_module, training_loop.py:116), whose
    reducer all-reduces a bucket as soon as autograd has delivered its gradients: out of one node they would all arrive together,
    behind the last launch of the pass.

    The flat gradient buffer fills from its end (engine.Tape.progress: a layer used late in the forward lies late in the buffer
    and is differentiated first), so it is cut into ``nseg`` ranges of equal size at parameter boundaries.  Node s runs the slice
    of the recorded backward that completes range s; the chain is built so that autograd reaches node nseg - 1 (which also returns
    the network output) first and node 0 (which ran the forward and takes x) last.  A node hands over the PREVIOUS slice's range:
    its weight-gradient launches were enqueued a slice ago on the gradient stream, so waiting for them (an event, not a join) does
    not hold up the input-gradient chain.  Node 0 delivers the last two ranges behind the pass's own final join.
    Plain torch.autograd only (no functorch transforms: those keep the single node)."""

    @staticmethod
    def forward(ctx, a, t, ctrl, idx, *params):
        ctx.ctrl, ctx.idx = ctrl, idx
        if idx == 0:
            eng = ctrl.net._get_engine()
            tape = Tape()
            if ctrl.holder.loss is not None:
                y, ctrl.holder.loss_sum = _forward_loss(eng, a, t, ctrl.dt, ctrl.holder.loss, tape)
            else:
                y = eng.forward(a, t, ctrl.dt, tape=tape, want_dx=True, forcing=ctrl.holder.forcing)
            ctrl.holder.tape = tape
            ctrl.y = y
            ctrl.x_needs_grad = bool(ctx.needs_input_grad[0]) and ctrl.holder.loss is None
        if idx == ctrl.nseg - 1:
            y, ctrl.y = ctrl.y, None
            return y
        return torch.zeros((), dtype=torch.float32, device=ctrl.net._get_engine().flat.device)  # the token the next node hangs on

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from . import ops
        ctrl, s = ctx.ctrl, ctx.idx
        eng: Engine = ctrl.net._get_engine()
        lay = eng.layout
        tape = ctrl.holder.tape
        if s == ctrl.nseg - 1:  # autograd's first stop: start the pass
            m = tape.meta
            dt = m["dt"]
            g_nhwc = torch.empty((m["B"] * m["H"] * m["W"], lay.cout_pad), dtype=TORCH_DTYPE[dt], device=g.device)
            if "loss" in m:
                _loss_output_gradient(m, g, g_nhwc, lay, dt)
            else:
                ops.nchw_to_nhwc(g.contiguous().float(), None, None, g_nhwc, m["B"], m["C"], m["H"] * m["W"], lay.cout_pad, dt)
            ctrl.fg = torch.zeros_like(eng.flat)  # gradients of this call only (autograd accumulates into .grad itself)
            ctrl.it = eng.backward_steps(tape, g_nhwc, want_dx=ctrl.x_needs_grad, want_dw=True)
        dx = None
        saved, eng.flat_grad = eng.flat_grad, ctrl.fg
        try:
            try:
                low = lay.numel
                while s == 0 or low > ctrl.bounds[s]:
                    low = next(ctrl.it)
            except StopIteration as done:
                dx = done.value
        finally:
            eng.flat_grad = saved
        side = eng.grad_stream()
        if side is not None and s > 0:
            ev = torch.cuda.Event()
            ev.record(side)
            ctrl.events[s] = ev
            prev = ctrl.events.pop(s + 1, None)
            if prev is not None:  # the range handed over now was enqueued a slice ago: this wait is (nearly) free
                torch.cuda.current_stream().wait_event(prev)
        fg = ctrl.fg
        views = lay.views
        grads = tuple(torch.as_strided(fg, views[k][1], views[k][2], views[k][0]) for k in ctrl.owned[s])
        if s == 0:
            ctrl.it = ctrl.fg = None
            return (dx, None, None, None) + grads
        return (torch.zeros((), dtype=torch.float32, device=fg.device), None, None, None) + grads


def _segmented_call(net, eng: Engine, x, t, dt: int, holder: "_TapeHolder", nseg: int):
    """Build the _GradSegment chain for one network call; returns its output (the network output or the fused loss tensor)."""
    lay = eng.layout
    names = list(lay.views)
    offs = [lay.views[k][0] for k in names]
    nseg = max(1, min(nseg, len(names)))
    bounds = [0]
    for i in range(1, nseg):  # range i starts at the parameter boundary nearest to i / nseg of the buffer (strictly increasing)
        target = i * lay.numel // nseg
        b = min((o for o in offs if o > bounds[-1]), key=lambda o: abs(o - target), default=None)
        if b is None:
            break
        bounds.append(b)
    nseg = len(bounds)
    seg_of = lambda off: max(i for i in range(nseg) if bounds[i] <= off)  # noqa: E731
    rng = [[k for k, o in zip(names, offs) if seg_of(o) == i] for i in range(nseg)]
    # node s hands over range s + 1 (node 0: ranges 0 and 1; the node autograd reaches first: nothing)
    owned = [rng[0] + (rng[1] if nseg > 1 else [])] + [rng[s + 1] if s + 1 < nseg else [] for s in range(1, nseg)]
    by_name = dict(zip(names, eng._bound))
    ctrl = _SegCtrl(net, holder, dt, nseg, bounds, owned)
    out = _GradSegment.apply(x, t, ctrl, 0, *[by_name[k] for k in owned[0]])
    for s in range(1, nseg):
        out = _GradSegment.apply(out, None, ctrl, s, *[by_name[k] for k in owned[s]])
    return out


def _forward_loss(eng: Engine, x, t, dt: int, req: dict, tape: Optional[Tape]):
    """The training-side composition of the reference in one launch sequence (src/thor/pipelines.py:22-35 around model/score.py:59-70):
    x_t = mu(t) x + sigma(t) eps fused into the input conversion, the network, and the unreduced loss (eps_pred - eps)^2 written
    straight from the network's NHWC output rows.  ``req["eps"]``: the seed of the regenerated noise stream, or the noise tensor.
    Returns (loss tensor (B,C,H,W) fp32, 1-element sum of it)."""
    from . import ops
    lay = eng.layout
    B, C, H, W = x.shape
    dev = x.device
    eps = req["eps"]
    tt = t.reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
    if tt.numel() != B:
        raise ValueError("the loss draws one t per batch item (src/thor/pipelines.py:29)")
    musig = torch.empty((B, 2), dtype=torch.float32, device=dev)
    ops.mu_sigma(tt, musig, B, float(req["eta"]))
    y = eng.forward(x, tt, dt, tape=tape, noise=(eps, musig), nhwc_out=True)
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=dev)
    loss_sum = torch.zeros(1, dtype=torch.float32, device=dev)
    if not ops.sq_err(y, eps, out, loss_sum, B, C, H * W, lay.cout_pad, dt):  # shape outside the fused kernel: layout pass + tensor arithmetic
        yn = torch.empty_like(out)
        ops.nhwc_to_nchw(y, yn, B, C, H * W, lay.cout_pad, dt)
        out = (yn - _materialize_eps(eps, out)) ** 2
        loss_sum = out.sum().reshape(1)
    if tape is not None:
        tape.meta["loss"] = dict(y=y, eps=eps)
    return out, loss_sum


def _materialize_eps(eps, like: torch.Tensor) -> torch.Tensor:
    if not isinstance(eps, int):
        return eps
    from . import ops
    e = torch.empty_like(like)
    ops.philox_normal(e, e.numel(), eps)
    return e


def _loss_output_gradient(m: dict, gy: torch.Tensor, g_nhwc: torch.Tensor, lay, dt: int) -> None:
    """d/dy of the unreduced loss (y - eps)^2 given the gradient ``gy`` of that tensor: g_nhwc = 2 (y - eps) gy.  ``.mean()`` (and a
    GradScaler's scale on top) hands back ONE value broadcast over the tensor (stride 0: _LossTensor.mean below keeps it that way,
    torch's own mean_backward materialises it): the fused loss-gradient kernel then reads the factor from device memory and eps is
    regenerated -- no host synchronisation, no pass over ``gy``.  Any other gradient takes tensor arithmetic."""
    from . import ops
    B, C, H, W = m["B"], m["C"], m["H"], m["W"]
    y, eps = m["loss"]["y"], m["loss"]["eps"]
    dummy = torch.zeros(1, dtype=torch.float32, device=gy.device)
    if gy.dtype == torch.float32 and all(s == 0 for s in gy.stride()):
        if isinstance(eps, int):
            if ops.mse_loss_grad_noise(y, eps, g_nhwc, dummy, B, C, H * W, lay.cout_pad, 2.0, dt, scaler=gy):
                return
            eps = _materialize_eps(eps, torch.empty((B, C, H, W), dtype=torch.float32, device=gy.device))
        ops.mse_loss_grad(y, eps.contiguous(), g_nhwc, dummy, B, C, H * W, lay.cout_pad, 2.0, dt, scaler=gy)
        return
    yn = torch.empty((B, C, H, W), dtype=torch.float32, device=gy.device)
    ops.nhwc_to_nchw(y, yn, B, C, H * W, lay.cout_pad, dt)
    g = (2.0 * (yn - _materialize_eps(eps, yn)) * gy).contiguous().float()
    ops.nchw_to_nhwc(g, None, None, g_nhwc, B, C, H * W, lay.cout_pad, dt)


class _MeanOfLoss(torch.autograd.Function):
    """mean() of the loss tensor from the sum its kernel already accumulated; the gradient goes back as ONE value broadcast over the
    tensor (an expanded view), where ``mean_backward`` would write -- and the loss node would have to read -- all B*C*H*W copies."""

    @staticmethod
    def forward(ctx, sq, loss_sum):
        ctx.shape = sq.shape
        return (loss_sum / sq.numel()).reshape(())

    @staticmethod
    def backward(ctx, g):
        n = 1
        for s in ctx.shape:
            n *= s
        return (g / n).reshape((1,) * len(ctx.shape)).expand(ctx.shape), None


class _LossTensor(torch.Tensor):
    """What the fused ``SDAPipeline.loss`` returns: the (B,C,H,W) tensor of squared errors, a plain tensor in every respect except
    that a full ``.mean()`` (training_loop.py:377) is answered by _MeanOfLoss.  Every other operation sees an ordinary tensor."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (torch.Tensor.mean, torch.mean) and len(args) == 1 and not kwargs and isinstance(args[0], _LossTensor):
            ls = getattr(args[0], "_c2w_loss_sum", None)
            if ls is not None:
                return _as_loss_scalar(_MeanOfLoss.apply(args[0].as_subclass(torch.Tensor), ls), getattr(args[0], "_c2w_engine", None))
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


class _LossScalar(torch.Tensor):
    """The mean of the fused loss and what the reference's loop derives from it before reading it back (training_loop.py:377,385:
    ``.mean().mul(loss_scaling)`` ... ``loss.detach().item()`` after ``optimizer.step()``).  ``item()`` on a plain tensor waits for
    everything enqueued on the stream -- the backward pass and the optimizer step, for a value that was final at the end of the
    forward -- and the chip then idles while the host prepares the next step (1.1 ms of a 49-ms step, profiles/r04_experiments.md
    section 1).  Here the value is PUBLISHED when it is produced: a one-thread launch behind its producer copies it into pinned host
    memory with a sequence number (Engine.publish), and ``item()`` / ``float()`` poll that memory -- the same bits, no stream
    synchronised, no second stream involved.  ``mul`` / ``div`` by a number stay in the class (published again), ``detach`` keeps
    its source's publication; every other operation sees, and returns, an ordinary tensor."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _SCALAR_READ and len(args) == 1 and not kwargs and isinstance(args[0], _LossScalar):
            v = _read_published(args[0])
            if v is not None:
                return v
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        if type(out) is torch.Tensor and out.dim() == 0 and out.is_cuda and args and isinstance(args[0], _LossScalar):
            src = args[0]
            eng = getattr(src, "_c2w_engine", None)
            if func in _SCALAR_ALIAS:  # no launch: the same value
                return _as_loss_scalar(out, eng, getattr(src, "_c2w_pub", None))
            if func in _SCALAR_KEEP and not any(isinstance(a, torch.Tensor) for a in args[1:]) and not kwargs:
                return _as_loss_scalar(out, eng)
        return out


_SCALAR_READ = (torch.Tensor.item, torch.Tensor.__float__)
_SCALAR_ALIAS = (torch.Tensor.detach, torch.detach)
_SCALAR_KEEP = (torch.Tensor.mul, torch.mul, torch.Tensor.__mul__, torch.Tensor.__rmul__, torch.Tensor.div, torch.div, torch.Tensor.__truediv__)


def _as_loss_scalar(t: torch.Tensor, eng, pub=None) -> torch.Tensor:
    if eng is None or not t.is_cuda or t.dim() != 0 or t.dtype != torch.float32 or os.environ.get("C2W_NO_EARLY_ITEM") == "1":  # the knob: A/B only
        return t
    if pub is None:  # produced by a launch just enqueued on the current stream: publish behind it
        pub = eng.publish(t.detach())
    s = t.as_subclass(_LossScalar)
    s._c2w_pub = pub
    s._c2w_engine = eng
    return s


def _read_published(s: "_LossScalar"):
    pub, eng = getattr(s, "_c2w_pub", None), getattr(s, "_c2w_engine", None)
    if pub is None or eng is None:
        return None
    return eng.published(*pub)


class ScoreUNet(torch.nn.Module):
    r"""U-Net score network on the MI355X engine.

    Arguments (as model/score.py:46): channels, embedding_dim, forcing_dim=0, **UNet kwargs
    (hidden_channels, hidden_blocks, attention_levels, kernel_size, activation, spatial, padding_mode).
    """

    def __init__(self, channels, embedding_dim, forcing_dim=0, **kwargs):
        super().__init__()
        # created FIRST, like the reference (model/score.py:49-51): the parameter creation order is the RNG order
        self.map_forcing = torch.nn.Linear(forcing_dim, embedding_dim) if forcing_dim > 0 else None
        self.embedding_dim = embedding_dim
        self.noise_features = 32
        self.unet = UNet(channels, channels, embedding_dim, **kwargs)
        self.map_layer0 = torch.nn.Linear(self.noise_features, embedding_dim)
        self.map_layer1 = torch.nn.Linear(embedding_dim, embedding_dim)
        self.precision = "auto"  # "auto" (the autocast dtype under torch.autocast, else fp32) | "fp32" | "bf16" | "fp16"
        self.ln_unbiased = True  # zuko.nn.LayerNorm uses torch.var_mean's default; see oracle/_shim/zuko/nn.py
        self.__dict__["_engine"] = None

    # ---- engine plumbing (kept out of state_dict / pickles / deep copies)
    def _get_engine(self) -> Engine:
        eng = self.__dict__.get("_engine")
        if eng is None:
            eng = Engine(self)
            self.__dict__["_engine"] = eng
        elif not eng.is_attached(self):
            eng.attach(self)  # parameters were replaced (.to(device), load via assign, dtype cast): re-flatten
        return eng

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_engine"] = None
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__["_engine"] = None

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k == "_engine" else copy.deepcopy(v, memo)
        return new

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """``nn.Module.load_state_dict`` that also takes the reference's own files: zuko's LayerNorm registers ``eps`` as a persistent
        buffer, so a reference ``training-state-*.ckpt`` / module ``state_dict()`` carries one ``*.eps`` key per LayerNorm (40 in the
        default network) next to the 228 parameter tensors (src/thor/checkpoint.py:37-57; SURVEY.md 8c iii) -- the LayerNorm here is
        parameter- and buffer-free, its epsilon a kernel argument -- and Fabric's module wrapper prefixes keys with
        ``_forward_module.`` on some versions.  Both are normalised away; everything else stays strict."""
        own = self.state_dict().keys() if any(k.endswith(".eps") for k in state_dict) else ()
        clean = type(state_dict)() if isinstance(state_dict, dict) else {}
        for k, v in state_dict.items():
            k = k[len("_forward_module."):] if k.startswith("_forward_module.") else k
            if k.endswith(".eps") and k not in own:
                if abs(float(v) - 1e-5) > 1e-12:
                    raise ValueError(f"{k} = {float(v)}: this engine's channel LayerNorm uses eps = 1e-5 (model/nn.py:44,154,183)")
                continue
            clean[k] = v
        return super().load_state_dict(clean, strict=strict, assign=assign)

    def compute_dtype(self) -> int:
        if self.precision == "fp32":
            return DTYPE_F32
        if self.precision == "bf16":
            return DTYPE_BF16
        if self.precision == "fp16":
            return DTYPE_F16
        if self.precision != "auto":
            raise ValueError(f"precision must be auto / fp32 / bf16 / fp16, got {self.precision!r}")
        if not torch.is_autocast_enabled():
            return DTYPE_F32
        return DTYPE_F16 if torch.get_autocast_dtype("cuda") == torch.float16 else DTYPE_BF16

    def forward(self, x: torch.Tensor, t: torch.Tensor, forcing: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert (forcing is None) or (self.map_forcing is not None)  # model/score.py:60
        if self.map_forcing is not None and forcing is None:
            raise ValueError("this network was built with forcing_dim > 0: forward() needs the forcing vector (model/score.py:65-66)")
        if forcing is not None and torch.is_grad_enabled() and forcing.requires_grad:
            raise NotImplementedError("gradients with respect to the forcing vector are not provided (no caller of the reference asks for them)")
        eng = self._get_engine()
        dt = self.compute_dtype()
        params = eng._bound  # the Parameter objects in layout order (is_attached has just verified they are the module's)
        req = self.__dict__.pop("_loss_request", None)  # set by SDAPipeline.loss around this call: return the unreduced loss instead
        if req is not None:
            if torch.is_grad_enabled() and any(p.requires_grad for p in params):
                holder = _TapeHolder(loss=req)
                nseg = self._segments(params, x)
                out = _segmented_call(self, eng, x, t, dt, holder, nseg) if nseg > 1 else _ScoreUNetFn.apply(x, t, self, dt, holder, *params)
                ls = holder.loss_sum
            else:
                out, ls = _forward_loss(eng, x, t, dt, req, None)
            out = out.as_subclass(_LossTensor)
            out._c2w_loss_sum = ls
            out._c2w_engine = eng
            return out
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        shape = x.shape
        x4 = x.reshape(-1, *shape[-3:]) if x.dim() != 4 else x
        nseg = self._segments(params, x4) if needs_grad else 1
        if nseg > 1:
            y = _segmented_call(self, eng, x4, t, dt, _TapeHolder(forcing=forcing), nseg)
        elif needs_grad or _in_functorch_transform(x):
            y = _ScoreUNetFn.apply(x4, t, self, dt, _TapeHolder(forcing=forcing), *params)
        else:
            y = eng.forward(x4, t, dt, forcing=forcing)
        return y.reshape(shape).to(x.dtype)

    # Parameter gradients delivered in this many segments during the backward pass (_GradSegment) instead of all at its end: None =
    # 8 when torch.distributed runs more than one rank (a DistributedDataParallel wrapper can then all-reduce its buckets under the
    # pass), else 1; C2W_GRAD_SEGMENTS or the attribute override it.
    grad_segments = None

    def _segments(self, params, x) -> int:
        n = self.grad_segments
        if n is None:
            env = os.environ.get("C2W_GRAD_SEGMENTS")
            if env is not None:
                n = int(env)
            else:
                import torch.distributed as dist
                n = 8 if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 else 1
        if n <= 1 or not torch.is_grad_enabled() or not all(p.requires_grad for p in params) or _in_functorch_transform(x):
            return 1
        return int(n)

    def _ordered_params(self, eng: Engine):
        named = dict(self.named_parameters())
        return [(k, named[k]) for k in eng.layout.views]


def _in_functorch_transform(x: torch.Tensor) -> bool:
    try:
        return torch._C._functorch.is_functorch_wrapped_tensor(x)
    except Exception:  # pragma: no cover
        return False
