"""Drop-in for the reference's optimizer seam: ``optimizer_kwargs.class_name = "torch.optim.AdamW"`` (train.py:175-180) is resolved by
the same ``construct_class_by_name`` as the network (training_loop.py:119-123: ``construct_class_by_name(params=net.parameters(),
**optimizer_kwargs)``), stepped at training_loop.py:380-384 after the loop has written ``g["lr"]`` into every param group.  Point the
string at ``climate2weather_amd.optim.AdamW``.

It is a ``torch.optim.Optimizer``: param groups, ``zero_grad``, ``state`` and ``state_dict`` in ``torch.optim.AdamW``'s layout (a
checkpoint written by either loads into the other), and torch's fused-optimizer AMP protocol (``_step_supports_amp_scaling``:
``GradScaler.step`` hands over ``grad_scale`` / ``found_inf`` as device tensors and the step is unscaled / skipped on the device, so
Fabric's "16-mixed" GradScaler runs without a 228-tensor ``unscale_`` and without a host synchronisation).

When a group's parameters are exactly those of ONE engine-backed ScoreUNet (the reference's case), a step is one launch of the fused
AdamW kernel over the network's flat fp32 buffer (engine.py::Layout; exp_avg / exp_avg_sq are two more flat buffers whose per-parameter
views populate ``state``), which also refreshes the 16-bit weight shadow the next forward reads -- instead of 228 tensors' worth of
multi-tensor launches followed by a separate cast.  Gradients that all exist but do not lie in one flat buffer (DDP bucket views) are
first copied into the engine's flat gradient buffer by one multi-tensor launch.  Anything else (other modules, CPU tensors, partial
parameter sets, missing gradients) takes torch's own functional AdamW per tensor on the same state: same numbers, torch's speed.
"""
from __future__ import annotations

import weakref
from typing import Dict, List, Optional

import torch

from . import ops
from .ops import DTYPE_F32


def _engine_of(p) -> Optional["object"]:
    from .engine import engine_of_parameter
    return engine_of_parameter(p)


_TORCH_DEFAULTS: Optional[dict] = None


def _torch_adamw_defaults() -> dict:
    global _TORCH_DEFAULTS
    if _TORCH_DEFAULTS is None:
        _TORCH_DEFAULTS = dict(torch.optim.AdamW([torch.zeros(1, requires_grad=True)]).defaults)
    return _TORCH_DEFAULTS


class AdamW(torch.optim.Optimizer):
    _step_supports_amp_scaling = True  # torch/amp/grad_scaler.py: step() is handed optimizer.grad_scale / optimizer.found_inf

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *, maximize=False, foreach=None,
                 capturable=False, differentiable=False, fused=None):
        if amsgrad or maximize or differentiable:
            raise ValueError("climate2weather_amd.optim.AdamW: amsgrad / maximize / differentiable are not supported (the reference uses none: train.py:176-181)")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 <= weight_decay:
            raise ValueError("invalid AdamW hyper-parameter")
        # the same keys, in the same order, as the installed torch.optim.AdamW's defaults (torch 2.1: ten keys; later versions add
        # decoupled_weight_decay=True): param_groups of a state_dict are interchangeable
        mine = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False, foreach=foreach,
                    capturable=capturable, differentiable=False, fused=fused)
        defaults = {k: mine.get(k, v) for k, v in _torch_adamw_defaults().items()}
        defaults.update({k: v for k, v in mine.items() if k not in defaults})
        super().__init__(params, defaults)
        self._flat: Dict[int, dict] = {}  # group index -> flat-path state (engine weakref, exp_avg / exp_avg_sq buffers, step counters)

    # ------------------------------------------------------------------ flat path
    def _plan(self, gi: int, group: dict) -> Optional[dict]:
        """Flat-path record of group ``gi`` if its parameters are exactly one attached engine's (checked against the engine's generation,
        i.e. re-checked only after a re-attach), else None."""
        params = group["params"]
        if not params or not params[0].is_cuda and not getattr(ops, "EMULATED", False):
            return None
        eng = _engine_of(params[0])
        if eng is None or eng.flat is None:
            return None
        st = self._flat.get(gi)
        if st is not None and st["eng"]() is eng and st["generation"] == eng.generation:
            return st
        bound = eng._bound
        if len(params) != len(bound) or {id(p) for p in params} != {id(p) for p in bound}:
            return None
        base = eng.flat.data_ptr()
        for p, (off, shape, strides) in zip(bound, eng.layout.views.values()):
            if p.dtype != torch.float32 or p.data_ptr() != base + 4 * off or tuple(p.stride()) != tuple(strides):
                return None
        old = st
        st = dict(eng=weakref.ref(eng), generation=eng.generation, m=torch.zeros_like(eng.flat), v=torch.zeros_like(eng.flat),
                  steps=0, amp=None, active=True)
        if old is not None and old["m"].shape == st["m"].shape and old["m"].device == st["m"].device:  # re-attached engine (e.g. load_state_dict with assign): keep the moments
            st["m"], st["v"], st["steps"], st["amp"] = old["m"], old["v"], old["steps"], old["amp"]
        else:  # moments torch-style state may already hold (load_state_dict before the first step, or steps taken on the per-tensor path)
            for p, (off, shape, strides) in zip(bound, eng.layout.views.values()):
                ps = self.state.get(p)
                if ps:
                    torch.as_strided(st["m"], shape, strides, off).copy_(ps["exp_avg"])
                    torch.as_strided(st["v"], shape, strides, off).copy_(ps["exp_avg_sq"])
                    st["steps"] = max(st["steps"], int(float(ps["step"])))
        step_t = torch.tensor(float(st["steps"]), dtype=torch.float32)
        st["step_t"] = step_t
        for p, (off, shape, strides) in zip(bound, eng.layout.views.values()):  # torch's per-parameter state: views of the flat moments
            self.state[p] = dict(step=step_t, exp_avg=torch.as_strided(st["m"], shape, strides, off),
                                 exp_avg_sq=torch.as_strided(st["v"], shape, strides, off))
        self._flat[gi] = st
        return st

    @staticmethod
    def _flat_grad(eng) -> Optional[torch.Tensor]:
        """The flat buffer every ``p.grad`` is a view of (what the engine's backward returns to autograd: one private buffer per
        backward, stolen by AccumulateGrad), or None (missing gradients, DDP bucket views, foreign tensors)."""
        bound = eng._bound
        g0 = bound[0].grad
        if g0 is None or g0.dtype != torch.float32 or g0.device != eng.flat.device:
            return None
        views = eng.layout.views
        off0 = next(iter(views.values()))[0]
        base = g0.data_ptr() - 4 * off0
        store = g0.untyped_storage()
        if base < store.data_ptr() or base + 4 * eng.layout.numel > store.data_ptr() + store.nbytes():
            return None
        for p, (off, shape, strides) in zip(bound, views.values()):
            g = p.grad
            if g is None or g.data_ptr() != base + 4 * off or tuple(g.stride()) != tuple(strides) or g.dtype != torch.float32:
                return None
        lead = (base - store.data_ptr()) // 4
        whole = torch.empty(0, dtype=torch.float32, device=g0.device).set_(store)
        return whole[lead: lead + eng.layout.numel]

    def _step_flat(self, group: dict, st: dict, eng, gflat: torch.Tensor, grad_scale, found_inf) -> None:
        n = eng.layout.numel
        eng.refresh_version()
        ver = eng._version()
        sdt = next((d for d, sh in eng.shadows.items() if eng._shadow_ver.get(d) == ver), None)  # the 16-bit copy the last forward read
        shadow = eng.shadows[sdt] if sdt is not None else None
        b1, b2 = group["betas"]
        amp = st["amp"]
        if grad_scale is not None or found_inf is not None:
            if amp is None:  # {scale, -, found_inf, steps taken}: the layout the kernel reads (c2w_adamw_ema_scaled)
                amp = st["amp"] = torch.zeros(4, dtype=torch.float32, device=eng.flat.device)
                amp[3] = float(st["steps"])
            if grad_scale is not None:
                amp[0:1].copy_(grad_scale.reshape(1), non_blocking=True)
            else:
                amp[0:1].fill_(1.0)
            if found_inf is not None:
                amp[2:3].copy_(found_inf.reshape(1), non_blocking=True)
        elif amp is not None:  # a loop that stopped using its GradScaler: every step is taken, nothing to unscale
            amp[0:1].fill_(1.0)
        st["steps"] += 1
        st["step_t"] += 1  # attempted steps on the host; under AMP the device counter amp[3] holds the steps actually taken
        ops.adamw_ema(eng.flat, gflat, st["m"], st["v"], None, shadow, n, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                      float(group["weight_decay"]), st["steps"], 0.0, 1.0, scaler=amp)
        if amp is not None:  # count the step unless it was skipped, clear found_inf (scale untouched: growth = backoff = 1)
            ops.grad_scaler_update(amp, 1.0, 1.0, 1 << 30)
        eng.weights_changed(shadow_fresh=sdt)
        if sdt is not None:
            eng.prefetch_backward_operands(sdt)  # the next backward's transposed / packed operands, on the gradient stream, next to the next forward

    @staticmethod
    def _gather_grads(eng) -> Optional[torch.Tensor]:
        """Gradients that exist for every parameter but do NOT lie in one flat buffer (torch DDP with gradient_as_bucket_view=True: views of
        the reducer's buckets; zero_grad(set_to_none=False) followed by accumulation into foreign tensors): copied, by one multi-tensor
        launch, into the engine's flat gradient buffer, so that the step stays the fused one.  None if a gradient is missing or not a
        dense fp32 tensor on the engine's device (then the step goes per tensor)."""
        bound = eng._bound
        grads = [p.grad for p in bound]
        dev = eng.flat.device
        if any(g is None or g.dtype != torch.float32 or g.device != dev or g.is_sparse or g.shape != p.shape for g, p in zip(grads, bound)):
            return None
        flat = eng.ensure_grad_buffer()
        views = [torch.as_strided(flat, shape, strides, off) for off, shape, strides in eng.layout.views.values()]
        if hasattr(torch, "_foreach_copy_"):
            torch._foreach_copy_(views, grads)
        else:  # older torch (the reference pins 2.1.2): per-tensor copies, still one fused step behind them
            for v, g in zip(views, grads):
                v.copy_(g)
        return flat

    def fused_path_active(self, gi: int = 0) -> bool:
        """Did the last step of group ``gi`` run as the one fused launch over the flat buffer?"""
        st = self._flat.get(gi)
        return st is not None and bool(st["active"])

    def steps_taken(self, gi: int = 0) -> int:
        """AdamW steps actually applied to group ``gi`` (skipped overflow steps excluded; synchronises under AMP)."""
        st = self._flat.get(gi)
        if st is None or not st["active"]:
            ps = self.state.get(self.param_groups[gi]["params"][0])
            return int(float(ps["step"])) if ps else 0
        return int(st["amp"][3].item()) if st["amp"] is not None else st["steps"]

    # ------------------------------------------------------------------ torch.optim.Optimizer API
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grad_scale = getattr(self, "grad_scale", None)
        found_inf = getattr(self, "found_inf", None)
        for gi, group in enumerate(self.param_groups):
            st = self._plan(gi, group)
            if st is not None:
                eng = st["eng"]()
                gflat = self._flat_grad(eng)
                if gflat is None:
                    gflat = self._gather_grads(eng)
                if gflat is not None:
                    if not st["active"]:
                        self._enter_flat(gi, st)
                    self._step_flat(group, st, eng, gflat, grad_scale, found_inf)
                    continue
                if st["active"]:
                    self._leave_flat(gi, st)
            self._step_per_tensor(group, grad_scale, found_inf)
        return loss

    def _leave_flat(self, gi: int, st: dict) -> None:
        """Some gradient is missing or foreign: continue per tensor ON THE SAME STATE.  The record stays (its flat exp_avg / exp_avg_sq
        buffers are what ``state`` holds views of -- torch's functional AdamW updates them in place), marked inactive; only the shared
        step counter becomes one tensor per parameter, as torch's per-tensor path increments each.  Nothing is reallocated, and the
        next step whose gradients are all there re-enters the fused path (_enter_flat)."""
        steps = self.steps_taken(gi)
        for p in self.param_groups[gi]["params"]:
            ps = self.state.get(p)
            if ps:
                ps["step"] = torch.tensor(float(steps), dtype=torch.float32)
        st["active"] = False

    def _enter_flat(self, gi: int, st: dict) -> None:
        """Back on the fused path after per-tensor steps: the moments never left the flat buffers; the step counter is the per-tensor
        path's (every parameter of a group has taken the same number of steps unless gradients were missing for some: the maximum)."""
        steps = 0
        for p in self.param_groups[gi]["params"]:
            ps = self.state.get(p)
            if ps:
                steps = max(steps, int(float(ps["step"])))
        st["steps"] = steps
        st["step_t"].fill_(float(steps))
        if st["amp"] is not None:
            st["amp"][3] = float(steps)
        for p in self.param_groups[gi]["params"]:
            ps = self.state.get(p)
            if ps:
                ps["step"] = st["step_t"]
        st["active"] = True

    def _step_per_tensor(self, group: dict, grad_scale, found_inf) -> None:
        from torch.optim.adamw import adamw
        if found_inf is not None and bool(found_inf.item() != 0):  # fallback path: GradScaler's own rule, one host sync
            return
        params, grads, m, v, steps = [], [], [], [], []
        for p in group["params"]:
            if p.grad is None:
                continue
            if p.grad.is_sparse:
                raise RuntimeError("AdamW does not support sparse gradients")
            ps = self.state[p]
            if not ps:
                ps["step"] = torch.tensor(0.0, dtype=torch.float32)
                ps["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                ps["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            params.append(p)
            grads.append(p.grad if grad_scale is None else p.grad / grad_scale.to(p.grad.device))
            m.append(ps["exp_avg"])
            v.append(ps["exp_avg_sq"])
            steps.append(ps["step"])
        if not params:
            return
        b1, b2 = group["betas"]
        adamw(params, grads, m, v, [], steps, foreach=group.get("foreach"), capturable=False, differentiable=False, fused=None, amsgrad=False,
              beta1=b1, beta2=b2, lr=group["lr"], weight_decay=group["weight_decay"], eps=group["eps"], maximize=False)
        for p in {id(_engine_of(q)): q for q in params if _engine_of(q) is not None}.values():
            _engine_of(p).refresh_version()  # torch wrote through the Parameter objects: their version counters moved, the caches follow

    def state_dict(self):
        """``torch.optim.AdamW.state_dict()``: per parameter ``step`` (a CPU scalar tensor), ``exp_avg``, ``exp_avg_sq`` -- each its own
        tensor (not views of the flat buffers: a loader may move or cast them one by one)."""
        sd = super().state_dict()
        taken = {}
        for gi, g in enumerate(sd["param_groups"]):
            if gi in self._flat and self._flat[gi]["active"]:
                for idx in g["params"]:
                    taken[idx] = self.steps_taken(gi)
        for idx, ps in sd["state"].items():
            ps = dict(ps)
            ps["step"] = torch.tensor(float(taken.get(idx, float(ps["step"]))), dtype=torch.float32)
            ps["exp_avg"], ps["exp_avg_sq"] = ps["exp_avg"].clone(), ps["exp_avg_sq"].clone()
            sd["state"][idx] = ps
        return sd

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)  # torch's own casting / device placement of every entry
        self._flat.clear()  # the flat buffers are rebuilt from ``state`` by the next step (_plan copies the loaded moments in)
