"""The legs of bench.py that are NOT the headline: per-launch timing of the implicit-GEMM kernels (by_kernel), the
reference-shaped loop on the drop-in path (module_api), BASELINE configs[4] (deep_variant) and configs[3] as the reference ships it
(sampler_configs3).  bench.py imports them, times the headline itself and prints the one compact line; the tools/ scripts call these
functions directly (`bench.module_api`, `bench.sampler_configs3` stay importable from bench)."""
import json
import os
import subprocess
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
BENCH_PY = os.path.join(REPO, "bench.py")

DEFAULT_CFG = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3,
                   padding_mode="zeros", attention_levels=[4])  # configs/sda_unet.yml
GFLOP_FWD = {65: 116.98, 52: 116.00}  # algorithmic GFLOP per (C,128,128) window forward
GFLOP_FWD_DEEP = 473.03  # C = 80, 256 x 256
MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 / fp16 (MI355X_MICROARCH.md)



# ----------------------------------------------------------------------------------------------------------------- kernel timing
class LaunchTimer:
    """HIP events around implicit-GEMM launches (ops.conv / ops.conv_wgrad), recorded on the stream each launch is enqueued on
    (torch's current stream at the call: the caller's stream for forward / input-gradient launches, the engine's gradient stream
    for weight gradients).  Launches are identified by the weight (or weight-gradient) pointer they are given, i.e. by LAYER, so
    the padded network-input / output convs are priced at their own algorithmic FLOP and not mistaken for a residual-block conv
    of the same padded geometry."""

    def __init__(self, ops, eng, dt, batch):
        self.ops, self.eng, self.dt, self.batch = ops, eng, dt, batch
        self.mode = "off"  # "off" | "dominant" | "all"
        self.events = []  # (layer, kind, geometry-dict, e0, e1[, layers of a grouped launch])
        self._conv, self._wgrad, self._wgrad_grouped = ops.conv, ops.conv_wgrad, ops.conv_wgrad_grouped
        self.fw, self.dg, self.gw = {}, {}, {}
        self.dominant = set()

    def index_layers(self):
        """pointer -> layer tables (after a warm-up step: every cached operand exists)."""
        eng, lay = self.eng, self.eng.layout
        from climate2weather_amd.ops import DTYPE_F32
        for rec in lay.convs.values():
            d = DTYPE_F32 if rec.lin else self.dt
            self.fw[eng._w(rec, d).data_ptr()] = rec
            if rec.dg_off >= 0:
                self.dg[eng._wT(rec, d).data_ptr()] = rec
            if eng.use_packed_weights and not rec.lin:  # the stage-major copies the 16x16-tile launches are handed instead
                pf = eng._packed("f", rec, d) if rec.name in eng._pk_want.get(("f", d), ()) else None  # only what the warm-up steps asked for
                if pf is not None:
                    self.fw[pf.data_ptr()] = rec
                pd = eng._packed("d", rec, d) if rec.name in eng._pk_want.get(("d", d), ()) else None
                if pd is not None:
                    self.dg[pd.data_ptr()] = rec
            self.gw[eng._gw(rec).data_ptr()] = rec
        for name, buf in eng._gwpad.items():
            self.gw[buf.data_ptr()] = lay.convs[name]
        lv0 = lay.levels[0]
        self.dominant = {n for n, r in lay.convs.items() if ".residue." in n and r.rows == lv0.channels and r.cin == lv0.channels
                         and (n.startswith("unet.descent.0.") or n.startswith(f"unet.ascent.{len(lay.levels) - 1}."))}

    def install(self):
        def conv(x, w, bias, y, g, dtype, **kw):
            rec = None
            if self.mode != "off":
                rec = self.fw.get(w.data_ptr()) or self.dg.get(w.data_ptr())
                if self.mode == "dominant" and (rec is None or rec.name not in self.dominant or g["B"] != self.batch):
                    rec = None
            if rec is None:
                return self._conv(x, w, bias, y, g, dtype, **kw)
            kind = "fwd" if w.data_ptr() in self.fw else "dgrad"
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._conv(x, w, bias, y, g, dtype, **kw)
            e1.record()
            self.events.append((rec, kind, g, e0, e1))

        def conv_wgrad(x, dy, dw, g, dtype, **kw):
            rec = self.gw.get(dw.data_ptr()) if self.mode == "all" else None
            if rec is None:
                return self._wgrad(x, dy, dw, g, dtype, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._wgrad(x, dy, dw, g, dtype, **kw)
            e1.record()
            self.events.append((rec, "wgrad", g, e0, e1))

        def conv_wgrad_grouped(items, g, dtype, **kw):  # the weight gradients of a level side in one launch (+ one reduction launch)
            rec = self.gw.get(items[0][2].data_ptr()) if self.mode == "all" else None
            if rec is None:
                return self._wgrad_grouped(items, g, dtype, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._wgrad_grouped(items, g, dtype, **kw)
            e1.record()
            self.events.append((rec, "wgrad", g, e0, e1, len(items)))
        self.ops.conv, self.ops.conv_wgrad, self.ops.conv_wgrad_grouped = conv, conv_wgrad, conv_wgrad_grouped

    def uninstall(self):
        self.ops.conv, self.ops.conv_wgrad, self.ops.conv_wgrad_grouped = self._conv, self._wgrad, self._wgrad_grouped

    @staticmethod
    def flop(rec, kind, g) -> float:
        """ALGORITHMIC FLOP of one launch: 2 x output pixels of the layer's forward x rows x taps x REAL input channels (padding
        channels of the network-input / output convs do no algorithmic work).  An input-gradient launch walks the forward's output
        grid (= its own input grid), a weight-gradient launch is handed the forward geometry."""
        pix = g["B"] * (g["Hin"] * g["Win"] if kind == "dgrad" else g["Hout"] * g["Wout"])
        return 2.0 * pix * rec.rows * rec.taps * rec.cin

    def family(self, rec, kind, g) -> str:
        lay = self.eng.layout
        n = rec.name
        if rec.lin:
            return f"{kind} linear fp32 (time MLP, modulation)"
        if n in ("unet." + lay.levels[0].head_key, "unet." + lay.levels[0].tail_key):
            return f"{kind} edge conv {rec.cin}->{rec.rows} (padded to 64-channel chunks) @{g['Hout'] if kind != 'dgrad' else g['Hin']}"
        if rec.taps == 1:
            return f"{kind} 1x1 {rec.cin}->{rec.rows} (attention qkv / proj)"
        if ".heads." in n:
            return {"fwd": "fwd 3x3 stride-2 (S2)", "dgrad": "dgrad of stride-2 (TS2)", "wgrad": "wgrad of stride-2 (S2)"}[kind] + f" {rec.cin}->{rec.rows}"
        side = g["Hin"] if kind == "dgrad" else g["Hout"]
        return f"{kind} 3x3 s1 {rec.cin}->{rec.rows} @{side}x{side}"

    def summarise(self, select=None, steps=1):
        """-> {family: dict(launches_per_step, avg_ms, ms_per_step, gflop_per_launch, tflops, frac)} over the recorded events."""
        fam = {}
        for ev in self.events:
            rec, kind, g, e0, e1 = ev[:5]
            layers = ev[5] if len(ev) > 5 else 1  # a grouped weight-gradient launch covers several layers of one shape
            if select is not None and not select(rec, kind):
                continue
            name = self.family(rec, kind, g) + (f" (grouped: {layers} layers per launch)" if layers > 1 else "")
            f = fam.setdefault(name, [0, 0.0, 0.0, 0])
            f[0] += 1
            f[1] += e0.elapsed_time(e1)
            f[2] += self.flop(rec, kind, g) * layers
            f[3] += layers
        out = {}
        for k, (n, ms, fl, nl) in fam.items():
            tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            peak = 157.0 if "fp32" in k else MFMA_PEAK_TFLOPS
            out[k] = dict(launches_per_step=round(n / steps, 2), avg_ms=round(ms / n, 4), ms_per_step=round(ms / steps, 3),
                          gflop_per_launch=round(fl / n / 1e9, 2), tflops=round(tf, 1), frac=round(tf / peak, 4))
            if nl != n:
                out[k]["layers_per_step"] = round(nl / steps, 2)
                out[k]["ms_per_layer"] = round(ms / nl, 4)
        return out


def _empty_cache():
    """Between the extra legs of this process the allocator's cached blocks go back to the driver (C2W_BENCH_KEEP_CACHE=1: keep them).
    Either way a leg that runs late in a long-lived process is 2-6 % slower than the same leg in a process of its own (round 4:
    module-API legs 0.93-0.95 of the headline after empty_cache(), 0.915-0.976 on a kept cache depending on what ran before,
    0.97 / 0.94 in a fresh process; the first conditioned-sampler leg 5.7 k -> 2.2 k window-forwards/s on a kept cache) -- memory handed
    back and re-obtained comes in fragments.  The legs that are compared WITH the headline (module_api) therefore run in a child
    process, like the headline's own ranks."""
    if os.environ.get("C2W_BENCH_KEEP_CACHE") != "1":
        torch.cuda.empty_cache()


def _step_stats(ms):
    """per-step GPU times between consecutive step-boundary events of rank 0 (ms_per_step above is the wall clock of the whole region)"""
    srt = sorted(ms)
    return dict(median=round(srt[len(srt) // 2], 3), min=round(srt[0], 3), max=round(srt[-1], 3), mean=round(sum(ms) / len(ms), 3), n=len(ms))



def module_api(dev, a, trainer_windows_per_s, legs=("bf16_autocast", "fp16_autocast_gradscaler", "trainer_fp16", "trainer_bf16", "trainer_bf16_c52", "trainer_bf16_b64"), item=True, lazy=False,
               wrap=None):
    """What a maintainer gets who changes ONLY the five class_name / func_name strings of train.py:164-193 and
    leaves training_loop.py alone: the loop of training_loop.py:369-391, statement for statement -- optimizer.zero_grad(); data =
    next(dataset_iterator) (a dense (B,C,H,W) tensor); loss = pipeline.loss(net, data).mean().mul(loss_scaling) under autocast;
    backward; lr written into the param groups; optimizer.step(); loss.item(); ema.update() -- on the default network at the
    benchmarked size.  Two legs: bf16 autocast, and the reference's own arithmetic (Fabric "16-mixed", train.py:98) = fp16 autocast +
    torch.amp.GradScaler stepping the optimizer.  Plus the fused Trainer in fp16 (device-resident loss scale) for comparison."""
    from climate2weather_amd.data import DeviceWindowFeed, SyntheticWindowDataset
    from climate2weather_amd.ema import StandardEMA
    from climate2weather_amd.lr import linear_learning_rate_schedule
    from climate2weather_amd.optim import AdamW
    from climate2weather_amd.pipelines import SDAPipeline
    from climate2weather_amd.score import ScoreUNet
    from climate2weather_amd.training import Trainer
    w = 2 * a.markov_order + 1
    C, B = a.vars * w, a.batch
    steps, warm = max(a.steps, 20), max(a.warmup, 3)  # 20 steps = 1 s per leg: the chip's clock wanders by +-3 % over half a second
    total_ndata = B * (steps + warm + 2) * 4
    res = dict(note="training_loop.py:369-391 with network / optimizer / pipeline / EMA / lr schedule resolved from this package's class names; B = %d, "
                    "C = %d, %dx%d; loss.item() every step as the reference does; vs_trainer = windows/s over the headline Trainer's" % (B, C, a.size, a.size),
               trainer_windows_per_s=trainer_windows_per_s)

    def timed(step, windows=B):
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):
            step()
            marks[i + 1].record()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        return dict(windows_per_s=round(windows / dt, 1), ms_per_step=round(1e3 * dt, 3), vs_trainer=round(windows / dt / trainer_windows_per_s, 4),
                    step_ms=_step_stats([marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]))

    for name, ac, use_scaler in (("bf16_autocast", torch.bfloat16, False), ("fp16_autocast_gradscaler", torch.float16, True)):
        if name not in legs:
            continue
        torch.manual_seed(0)
        net = ScoreUNet(channels=C, spatial=2, activation=torch.nn.SiLU, **DEFAULT_CFG).to(dev)
        net.train()
        mod = wrap(net) if wrap is not None else net  # tools/bench_module_api.py --ddp: torch's DistributedDataParallel (fabric.setup_module)
        pipeline = SDAPipeline()
        optimizer = AdamW(params=net.parameters(), lr=1e-4, weight_decay=1e-3, betas=[0.9, 0.999])  # train.py:176-181 through the class_name seam
        ema = StandardEMA(net=net)
        scaler = torch.amp.GradScaler("cuda") if use_scaler else None  # what Fabric's "16-mixed" precision plugin wraps backward / step in
        ds = SyntheticWindowDataset(n_frames=1024 + w - 1, n_vars=a.vars, height=a.size, width=a.size, window=w, seed=0)
        feed = DeviceWindowFeed(ds, dev, seed=0)
        state = dict(cur_ndata=0, losses=[])

        def step():
            optimizer.zero_grad()
            data = feed.next_batch(B, lazy=lazy)
            with torch.autocast("cuda", dtype=ac):
                loss = pipeline.loss(net=mod, x=data).mean().mul(1.0)
            (scaler.scale(loss) if scaler is not None else loss).backward()
            lr = linear_learning_rate_schedule(state["cur_ndata"], total_ndata, 1e-4)
            for g in optimizer.param_groups:
                g["lr"] = lr
            if scaler is not None:
                scaler.step(optimizer)
                scaler.update()
            else:
                optimizer.step()
            state["losses"].append(loss.detach().item() if item else loss.detach())
            state["cur_ndata"] += B
            ema.update(cur_ndata=state["cur_ndata"], batch_size=B)

        r = timed(step)
        r.update(final_loss=round(float(state["losses"][-1]), 5), flat_optimizer_path=optimizer.fused_path_active(), optimizer_steps_taken=optimizer.steps_taken(),
                 loss_scale=scaler.get_scale() if scaler is not None else None)
        res[name] = r
        del net, mod, optimizer, ema, feed, ds, pipeline, step  # the allocator keeps its blocks: the next leg has the same working set
    # the fused Trainer in the reference's arithmetic type (loss scale, inf check and skipped steps on the device)
    # ... and in bf16 on the reference's own recipe: 4 variables x window 13 = 52 channels (run_training.sh:39-45)
    # ... and at 64 windows per GPU: the reference's global batch of 512 (run_training.sh:43-45) strong-scaled over the 8 GPUs of a node
    for leg, prec, nvars, Bl in (("trainer_fp16", "fp16", a.vars, B), ("trainer_bf16", "bf16", a.vars, B), ("trainer_bf16_c52", "bf16", 4, B),
                                 ("trainer_bf16_b64", "bf16", a.vars, 64)):
        if leg not in legs:
            continue
        torch.manual_seed(0)
        net = ScoreUNet(channels=nvars * w, spatial=2, activation=torch.nn.SiLU, **DEFAULT_CFG).to(dev)
        tr = Trainer(net, SDAPipeline(), lr_fn=lambda n: linear_learning_rate_schedule(n, total_ndata, 1e-4), weight_decay=1e-3, ema_rates=[0.9999],
                     precision=prec, batch_size=Bl, seed=1000)
        ds = SyntheticWindowDataset(n_frames=1024 + w - 1, n_vars=nvars, height=a.size, width=a.size, window=w, seed=0)
        feed = DeviceWindowFeed(ds, dev, seed=0)
        r = timed(lambda: tr.step(feed.next_batch(Bl, lazy=True)), windows=Bl)
        r.update(optimizer_steps_taken=tr.optimizer_steps_taken(), loss_scale=tr.loss_scale(), channels=nvars * w, windows_per_step=Bl)
        if Bl != B and a.size == 128:  # whole-step matrix-core fraction at this batch (fwd + dgrad + wgrad minus the input conv's dgrad)
            gf = GFLOP_FWD.get(nvars * w, 116.0)
            r["mfma_frac_whole_step"] = round(r["windows_per_s"] * (3 * gf - 1.96) / 1e3 / MFMA_PEAK_TFLOPS, 4)
        res[leg] = r
        del tr, net, feed, ds
    # the like-for-like ratio: every leg over the fused bf16 Trainer timed in THIS process, minutes after the headline and on the
    # same allocator state (vs_trainer compares with the headline line, taken in another process at another moment)
    if "trainer_bf16" in res:
        for k, v in res.items():
            if isinstance(v, dict) and "windows_per_s" in v and v.get("windows_per_step", B) == B:
                v["vs_trainer_same_process"] = round(v["windows_per_s"] / res["trainer_bf16"]["windows_per_s"], 4)
    return res


def module_api_child(a, trainer_windows_per_s):
    """module_api() in a process of its own (this one stays alive and idle meanwhile, its cached memory returned): the same
    conditions the headline number was taken under.  The child is this file with --module-api-child; it prints one JSON line."""
    cmd = [sys.executable, BENCH_PY, "--module-api-child", str(trainer_windows_per_s), "--steps", str(a.steps), "--warmup", str(a.warmup),
           "--batch", str(a.batch), "--vars", str(a.vars), "--markov-order", str(a.markov_order), "--size", str(a.size)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return dict(error="module-api child failed", returncode=r.returncode, stderr_tail=r.stderr[-600:])
        res = json.loads(lines[-1])
        res["process"] = "child process of bench.py (fresh allocator, like the headline's own ranks)"
        return res
    except subprocess.TimeoutExpired:
        return dict(error="module-api child timed out")


def deep_variant(dev, B=32):
    """BASELINE configs[4]: 5 variables x 16 frames = 80 channels, 256x256 windows (473.03 GFLOP forward per window), fp16 -- the
    training step and the forward -- and its sampler step (k = 7 -> window 15 -> 75 channels: the reference's windows are odd,
    SURVEY.md section 0) as eager launches and as a hipGraph replay."""
    import contextlib, io
    from climate2weather_amd.pipelines import SDAPipeline
    from climate2weather_amd.score import ScoreUNet
    from climate2weather_amd.score_fn import BatchedScoreFunction
    from climate2weather_amd.training import Trainer
    res = dict(config="80 ch x 256x256, fp16, B=%d/GPU (BASELINE configs[4])" % B)
    torch.manual_seed(0)
    net = ScoreUNet(channels=80, spatial=2, activation=torch.nn.SiLU, **DEFAULT_CFG).to(dev)
    tr = Trainer(net, SDAPipeline(), lr=1e-4, precision="fp16", ema_rates=[0.9999], seed=1)
    x = torch.randn(B, 80, 256, 256, device=dev) * 0.5 + 0.5
    # four warm-up steps: on freshly returned memory the allocator needs a few steps before every block the two streams hold in turn
    # exists (seen twice in round 4 with two: 59 and 65 windows/s instead of 600+, every timed step waiting on hipMalloc)
    for _ in range(4):
        tr.step(x)
    torch.cuda.synchronize()
    n = 4
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(n):
        tr.step(x)
        marks[i + 1].record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    res["train_step_ms"] = _step_stats([marks[i].elapsed_time(marks[i + 1]) for i in range(n)])
    res["train_windows_per_s"] = round(B / dt, 1)
    res["train_model_tflops"] = round(B / dt * (3 * GFLOP_FWD_DEEP - 7.9) / 1e3, 1)
    res["train_mfma_frac"] = round(res["train_model_tflops"] / MFMA_PEAK_TFLOPS, 4)
    # its own by_kernel: every implicit-GEMM launch of one more step, streams serialised (the same pass as the headline's)
    from climate2weather_amd import ops as _ops
    kt = LaunchTimer(_ops, tr.eng, tr.dt, B)
    kt.install()
    try:
        kt.index_layers()
        prev = tr.eng.use_grad_stream
        tr.eng.use_grad_stream = False
        tr.step(x)
        torch.cuda.synchronize()
        kt.mode = "all"
        ts0 = time.perf_counter()
        tr.step(x)
        torch.cuda.synchronize()
        ser_ms = 1e3 * (time.perf_counter() - ts0)
        kt.mode = "off"
        tr.eng.use_grad_stream = prev
        fam = kt.summarise(steps=1)
        gemm_ms = sum(v["ms_per_step"] for v in fam.values())
        res["by_kernel"] = dict(serialised_step_ms=round(ser_ms, 2), implicit_gemm_ms_per_step=round(gemm_ms, 2),
                                everything_else_ms_per_step=round(ser_ms - gemm_ms, 2),
                                kernels=[dict(kernel=k, **v) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms_per_step"])][:24])
    finally:
        kt.uninstall()
    net.precision = "fp16"
    tt = torch.rand(B, device=dev)
    with torch.no_grad():
        for _ in range(2):
            net(x, tt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            net(x, tt)
        torch.cuda.synchronize()
    dtf = (time.perf_counter() - t0) / n
    res["forward_windows_per_s"] = round(B / dtf, 1)
    del tr, x
    _empty_cache()
    k, F, L = 7, 5, 47
    torch.manual_seed(0)
    net = ScoreUNet(channels=F * (2 * k + 1), spatial=2, activation=torch.nn.SiLU, **DEFAULT_CFG).to(dev).eval()
    net.precision = "fp16"
    pipe = SDAPipeline()
    for graph in (False, True):
        with contextlib.redirect_stdout(io.StringIO()):
            sf = BatchedScoreFunction(net, markov_order=k, batch_size=33, device=dev, noise_process=pipe)
            sf.use_graphs = graph
            noise = torch.randn(L, F, 256, 256, device=dev)
            pipe.sample(sf, noise, steps=2, show_progressbar=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe.sample(sf, noise, steps=6, show_progressbar=False)
            torch.cuda.synchronize()
        d = (time.perf_counter() - t0) / 6
        res["sampler_steps_per_s" + ("_hipgraph" if graph else "_eager")] = round(1 / d, 2)
        res["sampler_window_forwards_per_s" + ("_hipgraph" if graph else "_eager")] = round((L - 2 * k) / d, 1)
    return res


def sampler_configs3(dev, precision="bf16", lengths=(49, 121, 8737), corrections=(0, 2), steps=3, members=8, log=None):
    """BASELINE configs[3] as the reference runs it (exp/downscaling.py:208-265 with exp/configs/000_on-model-eval/s16_t6.yml and
    001_clim-downscaling/biased_climate_hadgem.yml): F = 4 variables, k = 6 (window 13 -> 52 channels), 128x128, window batches of
    128, CONDITIONED on A = AvgPool2d(16) o x[::6] with the shipped likelihood_std / likelihood_gamma (exact_grad = False), for the
    shipped trajectory lengths L = 49 / 121 / 8737 hours and corrections 0 (shipped) and 2 (src/thor/pipelines.py:52 code default
    is non-zero).  Per leg: sampler steps/s, window-forwards/s and the members/hour a 256-step run would give; plus `members`
    co-sampled members at L = 49 (their windows share the network batches).  Synthetic state and observation."""
    import contextlib, io
    from climate2weather_amd.pipelines import SDAPipeline
    from climate2weather_amd.score import ScoreUNet
    from climate2weather_amd.score_fn import BatchedScoreFunction, PoolStrideOperator
    F, k, H = 4, 6, 128
    w = 2 * k + 1
    torch.manual_seed(0)
    net = ScoreUNet(channels=F * w, spatial=2, activation=torch.nn.SiLU, **DEFAULT_CFG).to(dev).eval()
    net.precision = precision
    pipe = SDAPipeline()
    A = PoolStrideOperator(16, 6)
    std = torch.tensor([0.1692666615037876, 0.0425178630338289, 0.3268027589410125, 0.3268027589410125]).view(1, F, 1, 1)
    gamma = 0.0007196856730011522
    legs = []

    def leg(L, nmem, c, bsz=128, floor=0, into=None, graphs=False):
        shape = (L, F, H, H) if nmem == 1 else (nmem, L, F, H, H)
        g = torch.Generator(device=dev).manual_seed(L)
        truth = torch.randn((L, F, H, H), device=dev, generator=g) * 0.5 + 0.5
        with contextlib.redirect_stdout(io.StringIO()):
            sf = BatchedScoreFunction(net, markov_order=k, batch_size=bsz, device=dev, noise_process=pipe)
            sf.window_batch_floor = floor
            sf.use_graphs = graphs  # the network part of a score evaluation replayed from a hipGraph (score_fn._score_graphed)
            sf.condition_on(A=A, y=A(truth), std=std, gamma=gamma, exact_grad=False)
            assert sf._fused_guidance is not None
            del truth
            noise = torch.randn(shape, device=dev, generator=g)
            pipe.sample(sf, noise, steps=1, corrections=c, tau=0.5, device=dev, show_progressbar=False)
            torch.cuda.synchronize()
            n = steps
            while True:
                t0 = time.perf_counter()
                x = pipe.sample(sf, noise, steps=n, corrections=c, tau=0.5, device=dev, show_progressbar=False)
                torch.cuda.synchronize()
                d = (time.perf_counter() - t0) / n
                if n * d >= 0.25 or n >= 32:  # a 6-ms step timed over 3 steps moved by 15 % between runs: short legs get up to 32 steps
                    break
                n = min(32, max(n + 1, int(0.3 / d) + 1))
        assert bool(torch.isfinite(x).all())
        nwin = (L - w + 1) * nmem
        r = dict(L=L, members=nmem, corrections=c, timed_steps=n, windows_per_score_evaluation=nwin, sampler_steps_per_s=round(1 / d, 3),
                 ms_per_sampler_step=round(1e3 * d, 2), window_forwards_per_s=round(nwin * (1 + c) / d, 1),
                 members_per_hour_at_256_steps=round(nmem * 3600.0 / (256 * d), 2))
        if bsz != 128 or floor != 0:
            r.update(batch_size=bsz, window_batch_floor=floor)
        if graphs:
            r.update(hipgraph=True)
        (legs if into is None else into).append(r)
        if log is not None:
            log(r)

    for L in lengths:
        for c in corrections:
            leg(L, 1, c)
    if members > 1:
        for c in corrections:
            leg(49, members, c)
    # BASELINE configs[4]'s "hipGraph-captured sampler step" where launch latency could matter: the short trajectories, one member.  (A
    # sampler step at L = 49 is ~125 launches in 5.8 ms with the GPU busy 98.7 % of it, profiles/r04_sampler_l49_step_table.txt: the
    # replay removes host work, not device time.)
    graph_legs = []
    for L in (49, 121):
        if L in lengths:
            leg(L, 1, 0, into=graph_legs, graphs=True)
    # what the product default does with the reference's other shipped batch size (exp/configs: batch_size 32): score_fn.py::window_batch_floor
    floor_legs = []
    if 8737 in lengths:
        from climate2weather_amd.score_fn import BatchedScoreFunction as _B
        leg(8737, 1, 0, bsz=32, floor=0, into=floor_legs)
        leg(8737, 1, 0, bsz=32, floor=_B.window_batch_floor, into=floor_legs)
    return dict(hipgraph=dict(note="the same legs (one member, no corrector) with the network launches of a score evaluation replayed from a hipGraph",
                              legs=graph_legs),
                window_batch_floor=dict(note="L = 8737, batch_size = 32 (the reference's other shipped value): exactly 32 windows per network call "
                                             "(floor 0) against the product default (launches of at least `window_batch_floor` windows); the legs "
                                             "below run exactly 128 windows per call", legs=floor_legs),
                config="F=4, k=6, 52 ch x 128x128, %s, window batch 128, conditioned on AvgPool2d(16) o x[::6] (s16_t6.yml std / gamma, exact_grad=False), "
                       "%d timed sampler steps per leg after 1 warm-up step (legs shorter than 0.25 s are re-timed over up to 32 steps: `timed_steps`)" % (precision, steps), legs=legs)
