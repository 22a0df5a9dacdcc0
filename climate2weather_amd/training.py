"""The training step of the reference (training_loop.py:369-391) on the MI355X engine.

    zero_grad -> [accumulation rounds: loss = pipeline.loss(net, data).mean(); backward] -> lr -> AdamW -> EMA

One process per GPU (``torch.distributed``, backend "nccl" = RCCL on ROCm).  What differs from the reference below
the API:
  * noise process, network forward, loss, and the whole backward are the engine's HIP sequences (no autograd graph); the noise
    eps is a counter-based (Philox) stream of a per-step seed and the (B, C, H, W) fp32 noise tensor is never written or read
    (``fused_noise=False`` or an injected ``eps`` restore the tensor).  Where the output conv runs on the 16x16-tile kernel (round 6)
    the stream is generated ONCE, by the input conversion, which keeps it as half-precision NHWC rows; the loss tail is the output
    conv's epilogue and reads them back -- the step's noise is then the stream rounded to half precision, in x_t and in the loss
    alike, and the network's prediction is never written.  Elsewhere the input conversion and the loss kernel each regenerate
    the fp32 stream (rounds 1-5; ``C2W_NO_LOSS_FUSION=1`` everywhere);
  * gradients live in ONE flat fp32 buffer laid out in reverse finalisation order; while backward is still running,
    finished buckets of it are all-reduced over xGMI (replaces Lightning Fabric's DDP wrapper, training_loop.py:116,375-378)
    -- a sum; the 1/world_size mean is folded into the optimizer kernel.  Everything a bucket needs (the optional cast to the
    bf16 wire format, the collective, the stream-side wait for it, the cast back, the optional chased update) is enqueued on a
    COMMUNICATION stream of the trainer's own, ordered behind the bucket's last weight-gradient launch by an event; the compute
    stream never waits for a collective before the end of the backward (``_on_progress`` / ``_finish_allreduce``);
  * AdamW (train.py:176-181) + EMA (src/thor/ema.py:23-27) + the 16-bit weight shadow refresh are one fused kernel;
  * precision "fp16" (the reference's own: Fabric "16-mixed", train.py:98) trains under a dynamic loss scale with
    torch.cuda.amp.GradScaler's rule (init 2^16, x2 every 2000 clean steps, /2 and skip the step on inf/nan) -- kept in
    device memory and applied by the kernels themselves, so a skipped step costs no host synchronisation.
"""
from __future__ import annotations

import contextlib
import os
import re
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import ops
from .engine import Tape
from .ops import DTYPE_BF16, DTYPE_F16, DTYPE_F32, TORCH_DTYPE
from .pipelines import SDAPipeline


# C2W_EMULATE_COLLECTIVE_US=n (A/B runs with ONE rank, profiles/r06_experiments.md): n microseconds of spinning on the stream that waits for
# each bucket's collective (torch.cuda._sleep counts ~2.35 GHz cycles on MI355X, climate2weather_amd/streams.py)
_EMULATED_COLLECTIVE_CYCLES = int(float(os.environ.get("C2W_EMULATE_COLLECTIVE_US", "0") or 0) * 2350)


def _current_stream(device):
    """(indirection: the emulated stream test swaps the two stream accessors)"""
    return torch.cuda.current_stream(device)


def _stream_ctx(stream):
    return torch.cuda.stream(stream)


class Trainer:
    def __init__(self, net, pipeline: Optional[SDAPipeline] = None, *, lr: float = 1e-4, lr_fn: Optional[Callable[[int], float]] = None,
                 betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-3, ema_rates: Sequence[float] = (0.9999,),
                 precision: str = "bf16", batch_size: Optional[int] = None, loss_scaling: float = 1.0,
                 process_group=None, bucket_mb: float = 25.0, init_scale: float = 65536.0, growth_factor: float = 2.0,
                 backoff_factor: float = 0.5, growth_interval: int = 2000, fused_noise: bool = True, seed: Optional[int] = None,
                 allreduce_dtype: Optional[str] = None):
        self.net = net
        self.pipeline = pipeline or SDAPipeline()
        self.lr, self.lr_fn = lr, lr_fn
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.ema_rates = list(ema_rates)
        if precision not in ("fp32", "bf16", "fp16"):
            raise ValueError(f"precision must be fp32 / bf16 / fp16, got {precision!r}")
        self.dt = {"fp32": DTYPE_F32, "bf16": DTYPE_BF16, "fp16": DTYPE_F16}[precision]
        self.scaler_cfg = (float(growth_factor), float(backoff_factor), int(growth_interval))
        self.loss_scaling = loss_scaling
        self.fused_noise = fused_noise  # eps regenerated inside the kernels (Philox stream of a per-step seed) instead of randn_like(x)
        self.last_noise_seed: Optional[int] = None
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.sync_grads = self.world > 1 or (bool(os.environ.get("C2W_FORCE_DIST")) and dist.is_initialized())
        self.rank = dist.get_rank(process_group) if dist.is_available() and dist.is_initialized() else 0
        self.batch_size = batch_size  # global batch (items per optimizer step); None -> B_gpu * world
        self.cur_ndata = 0
        self.step_count = 0
        eng = net._get_engine()
        self.eng = eng
        eng.ensure_grad_buffer(net, bind=True)
        n = eng.layout.numel
        dev = eng.flat.device
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.ema_flats = [eng.flat.clone() for _ in self.ema_rates]
        self.loss_sum = torch.zeros(1, dtype=torch.float32, device=dev)
        # The step's random draws (t, the noise seed / eps) come from generators this trainer owns, seeded from (seed, rank) the way
        # the reference seeds each process (training_loop.py:49: set_random_seed(seed, rank)): ranks that were seeded identically by
        # their caller still see independent noise.  ``seed`` defaults to torch's initial seed, so torch.manual_seed(s) before the
        # construction reproduces a run.
        base = int(torch.initial_seed() if seed is None else seed)
        mixed = (base * 0x9E3779B97F4A7C15 + (self.rank + 1) * 0xD1B54A32D192ED03) & ((1 << 63) - 1)
        self.rng_cpu = torch.Generator().manual_seed(mixed)
        self.rng_dev = torch.Generator(device=dev).manual_seed(mixed)
        # dynamic loss scale {scale, growth tracker, found_inf, optimizer steps taken}: fp16 only (GradScaler is a no-op otherwise)
        self.scaler: Optional[torch.Tensor] = None
        if self.dt == DTYPE_F16:
            self.scaler = torch.zeros(4, dtype=torch.float32, device=dev)
            ops.grad_scaler_init(self.scaler, float(init_scale))
        if self.sync_grads:  # same initial weights everywhere (DDP's initial broadcast)
            dist.broadcast(eng.flat, src=0, group=self.pg)
            eng.weights_changed()
            for e in self.ema_flats:
                e.copy_(eng.flat)
        # all-reduce buckets over the flat gradient buffer, last bucket = first finalised
        bucket_mb = float(os.environ.get("C2W_BUCKET_MB", bucket_mb))  # env: diagnostic sweep
        per = max(int(bucket_mb * (1 << 20) // 4), 1)
        self.buckets: List[tuple] = []
        end = n
        while end > 0:
            start = max(0, end - per)
            self.buckets.append((start, end))
            end = start
        # Wire format of the gradient all-reduce: None / "fp32" = the flat fp32 gradients themselves (288 MB per step, what DDP moves
        # for the reference); "bf16" (or C2W_ALLREDUCE_DTYPE=bf16) = each finished bucket is cast to bfloat16, summed over the ranks
        # in bfloat16 and cast back: 144 MB on the xGMI links, gradients carry bf16's 8 mantissa bits across the sum (the
        # compression DDP's bf16_compress_hook applies) while weights, moments and the local gradients stay fp32.
        allreduce_dtype = os.environ.get("C2W_ALLREDUCE_DTYPE", allreduce_dtype) or "fp32"
        if allreduce_dtype not in ("fp32", "bf16"):
            raise ValueError(f"allreduce_dtype must be fp32 or bf16, got {allreduce_dtype!r}")
        self.wire = torch.empty(n, dtype=torch.bfloat16, device=dev) if (allreduce_dtype == "bf16" and self.sync_grads) else None
        self._works: list = []
        self._next_bucket = 0
        self._chase = None  # (lr, step) while the optimizer update chases the backward of the current round
        # Host run-ahead is bounded to ONE step: step k is enqueued while step k - 1 runs (no launch bubble at the step boundary), but
        # not before step k - 2 has finished -- further ahead, the blocks the gradient stream still holds (record_stream) are not
        # reusable yet and the caching allocator answers every request with a fresh, synchronising hipMalloc (measured: 210 ms/step).
        self._step_done: List[torch.cuda.Event] = []
        # Opt-in (C2W_CHASE_OPT=1 or the attribute): measured on one MI355X it buys nothing -- 50.8-50.9 ms per step either way; the step is
        # bound by the clock the chip holds under the MFMA load, and the update's HBM stream next to it lowers that clock further.
        self.chase_optimizer = os.environ.get("C2W_CHASE_OPT", "0") == "1"

    # ------------------------------------------------------------------ all-reduce and optimizer, both chasing the backward
    # The flat gradient buffer is laid out in reverse finalisation order, so what backward has finished is a growing SUFFIX.  Per
    # 25-MB bucket of it, as soon as it is final: (multi-GPU) sum it over the ranks with RCCL, then (opt-in) run the fused AdamW + EMA
    # + 16-bit shadow kernel on exactly that range.  The rest of the backward never reads the flat parameters again
    # (engine.Tape.progress), only its own copies of them.
    #
    # Streams.  The backward runs on ONE stream since round 5 (engine.use_grad_stream off): the weight gradients are written by the
    # compute stream itself.  Nothing a bucket needs may be enqueued there -- with the bf16 wire format or the chased update the
    # sequence contains a stream-side wait for the collective (work.wait()), and on the compute stream that wait would put every
    # bucket's all-reduce (25 MB over xGMI: ~0.1-0.2 ms at 8 GPUs) on the input-gradient chain's critical path.  So the whole
    # sequence goes to the communication stream (``_comm_stream``: the engine's side stream, on another hardware queue), which waits
    # for an event the compute stream records behind the bucket's last weight-gradient launch (comm.wait_stream(compute)); the compute
    # stream waits for the communication stream ONCE, behind the last bucket (``_finish_allreduce``).  With C2W_WGRAD_STREAM=1 (rounds
    # 1-4) the gradient stream the weight gradients are written on IS the communication stream and stream order is the dependency.
    # The plain fp32 wire has no wait inside the backward (async collectives, waited for at the end) and stays on the compute stream.
    def _comm_stream(self):
        """The stream bucket collectives (and what surrounds them) are enqueued on; None: the caller's (CPU tensors, or no collective
        and no second stream)."""
        if not self.eng.flat.is_cuda:
            return None
        gs = self.eng.grad_stream()
        if gs is not None:
            return gs
        if os.environ.get("C2W_COMM_ON_COMPUTE") == "1":  # A/B knob: round 5's behaviour (everything issued from the compute stream)
            return None
        # fp32 wire without the chased update: the collectives are asynchronous (RCCL runs them on its own stream behind an event of
        # the issuing stream) and nothing waits for them before _finish_allreduce -- issued from the compute stream itself they cost
        # no stream hop (one rank, same box: 46.70 against 46.81 ms per step through the communication stream).  Anything with a
        # stream-side wait inside the backward (bf16 wire: cast -> sum -> WAIT -> cast back; chased update: WAIT -> AdamW) goes to the
        # communication stream.  C2W_COMM_STREAM=1 sends the fp32 wire there as well.
        return self.eng.side_stream() if self._wants_comm_stream() else None

    def _wants_comm_stream(self) -> bool:
        """Does the per-bucket sequence of this step contain a stream-side wait (see _comm_stream)?"""
        return self.sync_grads and (self.wire is not None or self._chase is not None or os.environ.get("C2W_COMM_STREAM") == "1")

    def _on_progress(self, off: int) -> None:
        if not (self._next_bucket < len(self.buckets) and self.buckets[self._next_bucket][0] >= off):
            return
        comm = self._comm_stream()
        if comm is not None and self.eng.grad_stream() is None:
            # one-stream backward: the buckets below are final at THIS point of the compute stream
            comm.wait_stream(_current_stream(self.eng.flat.device))
        with (_stream_ctx(comm) if comm is not None else contextlib.nullcontext()):
            while self._next_bucket < len(self.buckets) and self.buckets[self._next_bucket][0] >= off:
                s, e = self.buckets[self._next_bucket]
                work = None
                if self.sync_grads and self.wire is None:
                    work = dist.all_reduce(self.eng.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                elif self.sync_grads:  # bf16 on the wire: cast, sum, cast back -- all ordered on the communication stream
                    wb = self.wire[s:e]
                    wb.copy_(self.eng.flat_grad[s:e])
                    work = dist.all_reduce(wb, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                    work.wait()  # the COMMUNICATION stream waits for RCCL's (a stream-side wait; gloo: the host); the copy back must see the sum
                    if _EMULATED_COLLECTIVE_CYCLES and wb.is_cuda:
                        # measurement knob (one-rank runs: RCCL's all-reduce over a single rank launches no kernel at all): the stream that
                        # waits for the collective is held for the time a 25-MB bucket takes over xGMI -- on the communication stream
                        # that time must not show in the step, on the compute stream (C2W_COMM_ON_COMPUTE=1) all of it does
                        torch.cuda._sleep(_EMULATED_COLLECTIVE_CYCLES)
                    self.eng.flat_grad[s:e].copy_(wb)
                    work = None
                if self._chase is not None:
                    if work is not None:
                        work.wait()  # again the communication stream, never the compute stream
                    self._update_range(s, e, *self._chase)
                elif work is not None:
                    self._works.append(work)
                self._next_bucket += 1

    def _finish_allreduce(self) -> None:
        """Behind the backward: the remaining buckets, then the ONE point where the compute stream waits for the collectives."""
        self._on_progress(0)
        comm = self._comm_stream()
        with (_stream_ctx(comm) if comm is not None else contextlib.nullcontext()):
            for w in self._works:
                w.wait()
        if comm is not None:
            _current_stream(self.eng.flat.device).wait_stream(comm)
        self._works.clear()
        self._next_bucket = 0

    def _update_range(self, s: int, e: int, lr: float, step: int) -> None:
        """Fused AdamW (train.py:176-181) + EMA (src/thor/ema.py:23-27) + 16-bit shadow refresh on flat[s:e], on the current stream."""
        eng = self.eng
        shadow = eng.shadows.get(self.dt) if self.dt != DTYPE_F32 else None
        ema = self.ema_flats[0][s:e] if self.ema_flats else None
        ops.adamw_ema(eng.flat[s:e], eng.flat_grad[s:e], self.m[s:e], self.v[s:e], ema, shadow[s:e] if shadow is not None else None, e - s,
                      float(lr), self.betas[0], self.betas[1], self.eps, self.weight_decay, step,
                      float(self.ema_rates[0]) if self.ema_rates else 0.0, 1.0 / self.world, scaler=self.scaler)

    # ------------------------------------------------------------------ one optimizer step
    def step(self, batches, t: Optional[torch.Tensor] = None, eps: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One optimizer step (``_step``).  C2W_MAIN_PRIORITY=-1 (A/B knob, read per call): the whole step runs on a high-priority HIP
        stream of the trainer's own, so that the forward / input-gradient chain wins CUs over the weight-gradient stream."""
        if os.environ.get("C2W_MAIN_PRIORITY") == "-1" and self.eng.flat.is_cuda:
            hp = self.__dict__.get("_hp_stream")
            if hp is None:
                hp = self.__dict__["_hp_stream"] = torch.cuda.Stream(device=self.eng.flat.device, priority=-1)
            cur = torch.cuda.current_stream()
            hp.wait_stream(cur)
            with torch.cuda.stream(hp):
                loss = self._step(batches, t, eps)
            cur.wait_stream(hp)
            return loss
        return self._step(batches, t, eps)

    def _step(self, batches, t: Optional[torch.Tensor] = None, eps: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``batches``: one (B,C,H,W) fp32 GPU tensor (or data.WindowBatch), or a list of them = gradient-accumulation rounds
        (training_loop.py:373-378: gradients of the rounds are summed, only the last round synchronises).
        ``t`` (B,) and ``eps`` (B,C,H,W) may be injected (tests); otherwise drawn as src/thor/pipelines.py:29-31 does.
        Returns the last round's loss (device scalar, like the value the reference logs)."""
        if not isinstance(batches, (list, tuple)):  # one tensor, or one data.WindowBatch (windows still inside the dataset array)
            batches = [batches]
        eng = self.eng
        if len(self._step_done) >= 2:
            self._step_done.pop(0).synchronize()
        eng.flat_grad.zero_()
        lr = self.lr_fn(self.cur_ndata) if self.lr_fn is not None else self.lr
        self.step_count += 1
        n = eng.layout.numel
        shadow = eng.shadow_for(self.dt) if self.dt != DTYPE_F32 else None  # exists (and is current) before the update writes into it
        # the update chases the backward unless the dynamic loss scale must first see the WHOLE gradient (fp16: inf/nan check)
        chase = self.chase_optimizer and self.scaler is None
        loss = None
        for r, x in enumerate(batches):
            last = r == len(batches) - 1
            self._chase = (lr, self.step_count) if (chase and last) else None
            loss = self._forward_backward(x, t if last or t is None else None, eps if last or eps is None else None,
                                          sync=last and (self.sync_grads or chase))
        if self.sync_grads or chase:
            self._finish_allreduce()
        self._chase = None
        if chase:
            eng.join_grad_stream()  # the last buckets' updates were enqueued after backward's own join
        else:
            if self.scaler is not None:  # after the all-reduce: inf/nan survive the sum, so every rank takes the same decision
                ops.grad_scaler_check(eng.flat_grad, n, self.scaler)
            self._update_range(0, n, lr, self.step_count)
            if self.scaler is not None:
                ops.grad_scaler_update(self.scaler, *self.scaler_cfg)
        for rate, e in zip(self.ema_rates[1:], self.ema_flats[1:]):
            ops.ema_update(e, eng.flat, n, float(rate))
        eng.weights_changed(shadow_fresh=self.dt if shadow is not None else None)
        eng.prefetch_backward_operands(self.dt)  # next step's input-gradient operands, next to its forward instead of in front of its backward
        B = sum(b.shape[0] for b in batches)
        self.cur_ndata += self.batch_size if self.batch_size is not None else B * self.world
        if eng.flat.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._step_done.append(ev)
        return loss

    def _forward_backward(self, x, t, eps, sync: bool) -> torch.Tensor:
        eng, lay = self.eng, self.eng.layout
        B, C, H, W = x.shape
        dev = x.device
        if t is None:
            t = torch.rand(B, dtype=torch.float32, device=dev, generator=self._rng_for(dev))
        seed = None
        if eps is None:
            if self.fused_noise and x.is_cuda:  # one 62-bit seed per round from the trainer's CPU generator (no device synchronisation)
                seed = int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64, generator=self.rng_cpu).item())
                self.last_noise_seed = seed
            else:
                eps = torch.randn(x.shape, dtype=x.dtype, device=dev, generator=self._rng_for(dev))
        t = t.reshape(-1).to(dev).float().contiguous()
        if eps is not None:
            eps = eps.contiguous()
        musig = torch.empty((B, 2), dtype=torch.float32, device=dev)
        ops.mu_sigma(t, musig, B, self.pipeline.eta)
        tape = Tape()
        if sync:
            tape.progress = self._on_progress
        self.loss_sum.zero_()
        n = B * C * H * W
        gs = 2.0 * self.loss_scaling / n
        # regenerated noise: the loss tail rides the output convolution's epilogue where that kernel exists (round 6: the prediction is
        # never written, the noise is generated once -- the input conversion keeps it as half-precision rows, and the step's noise IS
        # that rounded stream) -- eng.forward returns dY and says so in tape.meta
        lf = dict(sum=self.loss_sum, gscale=gs, scaler=self.scaler) if seed is not None else None
        y = eng.forward(x, t, self.dt, tape=tape, noise=(seed if seed is not None else eps, musig), nhwc_out=True, loss=lf)
        if tape.meta.get("loss_fused"):
            eng.backward(tape, y)
            return self.loss_sum[0] * (self.loss_scaling / n)
        dy = torch.empty_like(y)
        if seed is None or not ops.mse_loss_grad_noise(y, seed, dy, self.loss_sum, B, C, H * W, lay.cout_pad, gs, self.dt, scaler=self.scaler):
            if eps is None:  # shape outside the fused kernel: materialise the same stream
                eps = torch.empty(tuple(x.shape), dtype=torch.float32, device=dev)
                ops.philox_normal(eps, eps.numel(), seed)
            ops.mse_loss_grad(y, eps, dy, self.loss_sum, B, C, H * W, lay.cout_pad, gs, self.dt, scaler=self.scaler)
        eng.backward(tape, dy)
        return self.loss_sum[0] * (self.loss_scaling / n)

    def _rng_for(self, dev) -> torch.Generator:
        if self.rng_dev.device != torch.device(dev):  # batches on another device than the parameters' (not the product path)
            self.rng_dev = torch.Generator(device=dev).manual_seed(self.rng_cpu.initial_seed())
        return self.rng_dev

    def optimizer_steps_taken(self) -> int:
        """AdamW steps actually applied: under the fp16 loss scale, steps whose gradients overflowed were skipped on the
        device (this read synchronises; the training loop itself never does)."""
        return int(self.scaler[3].item()) if self.scaler is not None else self.step_count

    def loss_scale(self) -> float:
        return float(self.scaler[0].item()) if self.scaler is not None else 1.0

    # ------------------------------------------------------------------ EMA access / state
    def ema_state_dicts(self):
        """[(rate, state_dict)] with the reference's key names (snapshot export, training_loop.py:250-265)."""
        out = []
        for rate, flat in zip(self.ema_rates, self.ema_flats):
            sd = {}
            for name, (off, shape, strides) in self.eng.layout.views.items():
                sd[name] = torch.as_strided(flat, shape, strides, off).clone()
            out.append((rate, {k: sd[k] for k in self.net.state_dict().keys()}))
        return out

    def _per_param(self, flat: torch.Tensor):
        """name -> strided view of a flat per-parameter buffer, in ``net.named_parameters()`` order (= the order the
        reference hands parameters to its optimizer, train.py:176-181)."""
        views = self.eng.layout.views
        return [(n, torch.as_strided(flat, views[n][1], views[n][2], views[n][0])) for n, _ in self.net.named_parameters()]

    def optimizer_state_dict(self):
        """``torch.optim.AdamW.state_dict()`` layout: what the reference's checkpoint holds under "optimizer"
        (src/thor/checkpoint.py:13-35 saves ``optimizer`` through Fabric = its state_dict)."""
        ms, vs = self._per_param(self.m), self._per_param(self.v)
        state = {}
        steps = self.optimizer_steps_taken()
        if steps > 0:
            for i, ((_, m), (_, v)) in enumerate(zip(ms, vs)):
                state[i] = dict(step=torch.tensor(float(steps)), exp_avg=m.clone(), exp_avg_sq=v.clone())
        lr = self.lr_fn(self.cur_ndata) if self.lr_fn is not None else self.lr
        group = dict(lr=float(lr), betas=tuple(self.betas), eps=self.eps, weight_decay=self.weight_decay, amsgrad=False, maximize=False,
                     foreach=None, capturable=False, differentiable=False, fused=None, params=list(range(len(ms))))
        return dict(state=state, param_groups=[group])

    def load_optimizer_state_dict(self, osd):
        self.m.zero_()
        self.v.zero_()
        step = 0
        ms, vs = self._per_param(self.m), self._per_param(self.v)
        for i, st in osd["state"].items():
            ms[int(i)][1].copy_(st["exp_avg"])
            vs[int(i)][1].copy_(st["exp_avg_sq"])
            step = max(step, int(float(st["step"])))
        self.step_count = step
        if self.scaler is not None:
            self.scaler[3] = float(step)
        if osd.get("param_groups"):
            g = osd["param_groups"][0]
            self.betas, self.eps, self.weight_decay = tuple(g["betas"]), g["eps"], g["weight_decay"]

    def state_dict(self):
        """The reference's training-state checkpoint, key for key (src/thor/checkpoint.py:13-35 over the objects of
        training_loop.py:132-138): ``state`` (progress counters), ``net`` (228-key state_dict), ``pipeline`` (its
        ``__dict__``), ``optimizer`` (AdamW state_dict), ``ema`` (``StandardEMA.state_dict()``: rates + one state_dict each)."""
        return dict(state=dict(cur_ndata=self.cur_ndata, total_elapsed_time=getattr(self, "total_elapsed_time", 0)),
                    net={k: v.clone() for k, v in self.net.state_dict().items()},
                    pipeline=dict(eta=self.pipeline.eta),
                    optimizer=self.optimizer_state_dict(),
                    ema=dict(rates=list(self.ema_rates), emas=[sd for _, sd in self.ema_state_dicts()]))

    def load_state_dict(self, sd):
        self.net.load_state_dict(sd["net"])  # tolerates zuko's `*.eps` buffer keys and Fabric's `_forward_module.` prefix (score.py)
        self.eng = self.net._get_engine()
        self.eng.weights_changed()
        self.load_optimizer_state_dict(sd["optimizer"])
        self.cur_ndata = int(sd["state"]["cur_ndata"])
        self.total_elapsed_time = sd["state"].get("total_elapsed_time", 0)
        if sd.get("pipeline"):
            self.pipeline.eta = sd["pipeline"].get("eta", self.pipeline.eta)
        if sd.get("ema") is not None:
            views = self.eng.layout.views
            for flat, esd in zip(self.ema_flats, sd["ema"]["emas"]):
                for name, t in esd.items():
                    name = name[len("_forward_module."):] if name.startswith("_forward_module.") else name
                    if name not in views and name.endswith(".eps"):
                        continue  # zuko LayerNorm buffer of a reference checkpoint
                    off, shape, strides = views[name]
                    torch.as_strided(flat, shape, strides, off).copy_(t)


CKPT_RE = re.compile(r"training-state-(\d+).ckpt")


def save_checkpoint(trainer: Trainer, run_dir: str) -> str:
    """``training-state-{kdata:07d}.ckpt`` as training_loop.py:353-363 names it."""
    path = os.path.join(run_dir, f"training-state-{trainer.cur_ndata // 1000:07d}.ckpt")
    torch.save(trainer.state_dict(), path)
    return path


def load_latest_checkpoint(trainer: Trainer, run_dir: str) -> Optional[str]:
    """src/thor/checkpoint.py:61-79: pick the highest-numbered training-state file."""
    best = None
    for f in os.listdir(run_dir) if os.path.isdir(run_dir) else []:
        m = CKPT_RE.fullmatch(f)
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    if best is None:
        return None
    path = os.path.join(run_dir, best[1])
    trainer.load_state_dict(torch.load(path, map_location=trainer.eng.flat.device, weights_only=False))
    return path
