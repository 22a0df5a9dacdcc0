#!/usr/bin/env python3
"""Headline benchmark: UNet denoise steps/sec = training windows/s through the reference's training step
(training_loop.py:369-391: noise -> ScoreUNet fwd -> MSE -> bwd -> grad all-reduce -> AdamW -> EMA) on the
default configs/sda_unet.yml network, synthetic (B, F*w, 128, 128) fields, bf16 compute, one process per GPU.

    python bench.py --gpus N --steps K --warmup W
        N > 1 without a launcher: this process (which never touches a GPU) starts N rank processes itself -- what
        fabric.launch() does for the reference (train.py:93-100) -- and relays rank 0's JSON line;
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
        the ranks are the launcher's: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment.

Rank 0 prints ONE compact JSON line (< 6 KB: the headline, `roofline`, `cpu_baseline` and a few scalars -- compact_line() below;
tests/test_bench_launcher.py bounds its size) and writes everything else (`by_kernel`, the deep variant, the module-API legs, the
sampler legs) to gpurun_out/bench_extras.json, named in the line's `extras_file`.  `roofline` is measured live (HIP events on the launch stream around every launch of the
dominant kernel inside the timed region); `by_kernel` comes from extra steps AFTER the timed region in which every implicit-GEMM
launch is bracketed by events and the two backward streams are serialised (each kernel alone on the chip); `cpu_baseline` is the
CPU oracle (oracle/, a plain-PyTorch restatement of the same step) timed on this box's host cores, rank 0 at N=1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

DEFAULT_CFG = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3,
                   padding_mode="zeros", attention_levels=[4])  # configs/sda_unet.yml
GFLOP_FWD = {65: 116.98, 52: 116.00}  # algorithmic GFLOP per (C,128,128) window forward
GFLOP_FWD_DEEP = 473.03  # C = 80, 256 x 256
MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 / fp16 (MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=128, help="windows per GPU per optimizer step (run_training.sh: 128)")
    p.add_argument("--vars", type=int, default=5, help="physical variables F (north_star: 5; reference recipe: 4)")
    p.add_argument("--markov-order", type=int, default=6)
    p.add_argument("--size", type=int, default=128)
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"],
                   help="bf16 (default), fp16 (the reference's autocast type; dynamic loss scale on the device) or fp32 (parity mode)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--sample-steps", type=int, default=2, help="sampler steps timed after the headline run (0 = skip)")
    p.add_argument("--kernel-steps", type=int, default=2, help="extra steps with every implicit-GEMM launch timed (by_kernel; 0 = skip)")
    p.add_argument("--no-extras", action="store_true", help="skip by_kernel, the deep-variant leg and the sampler legs")
    p.add_argument("--module-api-child", type=float, default=None, metavar="TRAINER_WINDOWS_PER_S",
                   help="internal: run only the module_api legs (bench.py starts itself with this for a fresh process) and print their JSON")
    p.add_argument("--light-extras", action="store_true",
                   help="only the extras every rank of a multi-GPU job runs (vendor GEMM, by_kernel, sampler legs incl. the member-sharded one)")
    return p.parse_args()


# ----------------------------------------------------------------------------------------------------------------- launcher
def visible_gpus(run=subprocess.run):
    """GPUs this process tree may use, counted WITHOUT a HIP / HSA call in THIS process (the parent must stay GPU-free: it only
    starts the ranks).  First the KFD topology in sysfs (nodes with SIMDs; CPU nodes have simd_count 0), narrowed by the
    *_VISIBLE_DEVICES masks; where that is not readable (containers) a throw-away child process asks the runtime.  None if neither
    answers (then --gpus is trusted and a rank without a device fails)."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    n = None
    try:
        n = 0
        for node in sorted(os.listdir(root)):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        n = None
    if n is not None:
        for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            mask = os.environ.get(var)
            if mask is not None:
                n = min(n, len([m for m in mask.split(",") if m.strip() != ""]))
        return n
    try:  # a child may initialise whatever it likes; this process does not
        r = run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
    except (OSError, ValueError, IndexError, subprocess.SubprocessError):
        return None


def launch(a, popen=subprocess.Popen, count=visible_gpus, grace=15.0, straggler_grace=120.0) -> int:
    """`--gpus N` without a launcher's environment: one child process per GPU (what fabric.launch() does for the reference,
    train.py:93-100), rendezvous on 127.0.0.1, rank 0's stdout is relayed, everything else goes to stderr.  This parent makes no HIP
    call; the children are fresh interpreters.  A rank that dies takes the job down: the survivors (stuck in the rendezvous or in a
    collective) are killed `grace` seconds later and the exit status is the failure's.  A rank that is still running
    `straggler_grace` seconds after the FIRST rank has exited cleanly (every rank leaves the same barrier; one stuck in
    destroy_process_group or a collective would otherwise keep this loop spinning forever) is killed too and the job fails."""
    import threading
    visible = count()
    if visible is not None and visible < a.gpus:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but only {visible} GPU(s) are visible; refusing to report fewer ranks than asked for\n")
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                           stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].communicate()[0]), daemon=True)  # drains rank 0's pipe while we watch
    reader.start()
    deadline, failed_armed, killed = None, False, False
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        if not failed_armed and any(rc not in (None, 0) for rc in rcs):
            failed_armed = True
            deadline = time.time() + grace if deadline is None else min(deadline, time.time() + grace)
        elif deadline is None and any(rc == 0 for rc in rcs):
            deadline = time.time() + straggler_grace
        if deadline is not None and time.time() >= deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
                    killed = True
            deadline = float("inf")
        time.sleep(0.05)
    reader.join(timeout=30)
    sys.stdout.write((out0[0] if out0 and out0[0] else b"").decode())
    sys.stdout.flush()
    rc = max(abs(rc) for rc in rcs)
    return rc if rc or not killed else 1  # a straggler that had to be killed is a failure even if every other rank exited 0


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


def pmc_traffic(a):
    """roofline.traffic: HBM bytes per forward launch of the dominant layers, from PMC passes over this same training step
    (tools/pmc_step.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES in separate runs, FETCH_SIZE
    doubled for gfx950; tools/pmc_step_summary.py -> profiles/r0N_pmc_step.json).  PMC counters cannot be read from inside this
    process, so the figure is a committed profile's -- and only if that profile was taken with a library built from the SAME sources
    as the one loaded now (the summary records c2w_sources_sha256); for any other build, workload or round it is null."""
    from climate2weather_amd import build as c2w_build
    mine = c2w_build.embedded_digest()
    prof, path = None, None
    for cand in sorted((f for f in os.listdir(os.path.join(REPO, "profiles")) if f.endswith("_pmc_step.json")), reverse=True):
        try:
            pr = json.load(open(os.path.join(REPO, "profiles", cand)))
        except (OSError, ValueError):
            continue
        if pr.get("c2w_sources_sha256") == mine and mine is not None:
            prof, path = pr, "profiles/" + cand
            break
    if prof is None or not (a.size == 128 and a.batch == 128 and a.precision == "bf16" and a.vars == 5):
        return dict(traffic=None, from_profile=None)
    try:
        fd = prof["forward_dominant_launches"]
        lnf = next((e for e in prof["kernels"] if e["kernel"].startswith("conv_patch_t3_kernel<16, unsigned short, 8, 2") and e["workgroups"] == 8192
                    and e.get("mfma_busy") is not None), {})
    except (KeyError, ValueError):
        return dict(traffic=None, from_profile=None)
    return dict(traffic=fd["hbm_bytes"],
                from_profile=dict(note="PMC passes over this training step on another box, library built from the same sources (tools/pmc_step.sh; rocprofv3 "
                                       "--kernel-trace --pmc, one counter group per run; FETCH_SIZE x 2 for gfx950 + WRITE_SIZE): mean over the forward launches of "
                                       "8192 workgroups with the epilogues the step runs (SiLU pair: 537 MB in + 2 x 537 MB out algorithmic; residual + LayerNorm "
                                       "emission: 2 x 537 MB in + 2 x 537 MB out)",
                                  source=path, c2w_sources_sha256=mine, traffic_bytes_per_launch=fd["hbm_bytes"], hbm_read_bytes=fd["hbm_read_bytes"],
                                  hbm_write_bytes=fd["hbm_write_bytes"], launches_profiled=fd["launches"],
                                  mfma_busy_layernorm_emission_flavour=lnf.get("mfma_busy"), clock_ghz_under_load=lnf.get("clock_ghz"),
                                  mfma_busy_by_kernel_in_step={f"{e['kernel']} x{e['workgroups']} workgroups": dict(
                                      mfma_busy=e.get("mfma_busy"), clock_ghz=e.get("clock_ghz"), launches=e.get("launches_mfma"))
                                      for e in prof["kernels"] if e.get("mfma_busy") is not None and e.get("launches_mfma", 0) >= 10},
                                  mfma_busy_note="SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the share of the cycles of the clock the chip "
                                                 "HOLDS in which the matrix pipe is busy -- rocprof's MFMA utilisation; `frac` above is FLOP/s against the 2.4 GHz spec peak"))


# ----------------------------------------------------------------------------------------------------------------- CPU baseline
def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(C, size, cfg):
    """the oracle (CPU restatement of the reference step) on this box's host cores -- 3 warm-up + 5 timed
    iterations (median), forward (eval, no_grad, B = 2) and fwd+bwd of `loss(net, x).mean()` (B = 2), at C = 52 (the reference's
    recipe) and the benchmarked C, at the thread count that maximises windows/s (swept first; all cores oversubscribe this size)."""
    from oracle import diffusion as od
    from oracle import unet as ou
    from climate2weather_amd.score import ScoreUNet
    B = 2
    ncpu = os.cpu_count() or 1
    g = torch.Generator().manual_seed(0)

    def make(Cc):
        torch.manual_seed(0)
        net = ScoreUNet(channels=Cc, spatial=2, activation=torch.nn.SiLU, **cfg)
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
        x = torch.randn(B, Cc, size, size, generator=g) * 0.5 + 0.5
        fwd = lambda a, b: ou.score_unet_forward(sd, a, b, cfg["hidden_blocks"], cfg["attention_levels"])
        return sd, x, fwd

    def run_fwd(sd, x, fwd):
        with torch.no_grad():
            fwd(x, torch.rand(B, generator=g))

    def run_train(sd, x, fwd):
        t = torch.rand(B, 1, 1, 1, generator=g)
        eps = torch.randn(x.shape, generator=g)
        loss = od.loss(fwd, x, t, eps).mean()
        torch.autograd.grad(loss, list(sd.values()))

    def timed(fn, args, warm, n):
        for _ in range(warm):
            fn(*args)
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn(*args)
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    main = make(C)
    sweep = {}
    for nt in sorted({n for n in (4, 8, 16, 32, 64, 128, ncpu) if n <= ncpu}):
        torch.set_num_threads(nt)
        sweep[nt] = round(B / timed(run_fwd, main, 1, 2), 2)
        if sweep[nt] < 0.5 * max(sweep.values()):  # past the knee: more threads only oversubscribe this size (256 threads: 60 s per pass)
            break
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    legs = {}
    for Cc in sorted({52, C}):
        m = main if Cc == C else make(Cc)
        tf, tt = timed(run_fwd, m, 3, 5), timed(run_train, m, 3, 5)
        legs[f"C={Cc}"] = dict(forward_windows_per_s=round(B / tf, 3), forward_ms_per_iter=round(1e3 * tf, 1),
                              train_windows_per_s=round(B / tt, 3), train_ms_per_iter=round(1e3 * tt, 1))
    return dict(value=legs[f"C={C}"]["train_windows_per_s"], unit="windows/s", cores=best, kind="port",
                sample=f"oracle (plain PyTorch CPU fp32 restatement) fwd+bwd of the same training step, B={B}, C={C}, {size}x{size}, 3 warm-up + 5 "
                       f"timed iterations (median), {best} threads = the best of the sweep below ({ncpu} logical CPUs on the box)",
                cpu_model=_cpu_model(), logical_cpus=ncpu, thread_sweep_forward_windows_per_s=sweep, legs=legs)


def _leg(out, name, fn):
    """Run one extra leg (everything outside the timed headline region).  A leg that raises must not cost the run its record: the
    exception goes into `errors` (extras file + a count in the compact line) and the headline line is still printed."""
    try:
        return fn()
    except Exception as e:  # noqa: BLE001 -- any failure of an extra leg is reported, never fatal
        import traceback
        sys.stderr.write(f"bench.py: extra leg {name!r} failed:\n{traceback.format_exc()}\n")
        if out is not None:
            out.setdefault("errors", {})[name] = f"{type(e).__name__}: {e}"[:400]
        return None


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



# ----------------------------------------------------------------------------------------------------------------- the line
COMPACT_LIMIT = 6144  # bytes; the driver keeps the last 8 KB of stdout (round 4: a 24 KB line left it nothing to parse)


def extras_path() -> str:
    return os.environ.get("C2W_BENCH_EXTRAS", os.path.join(REPO, "gpurun_out", "bench_extras.json"))


def _get(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def compact_line(full: dict, extras_file=None) -> str:
    """The ONE line stdout carries: the contract's keys, `roofline` and `cpu_baseline` reduced to their numbers and one short sentence
    each, and the scalars the other legs boil down to.  Everything it leaves out is in `extras_file`.  Never longer than COMPACT_LIMIT."""
    roof, cpu = full.get("roofline"), full.get("cpu_baseline")
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data", "config")}
    if roof:
        prof = roof.get("from_profile") or {}
        dom = next((v for k, v in (prof.get("mfma_busy_by_kernel_in_step") or {}).items() if ", 4, true> x8192" in k), {})
        line["roofline"] = dict(
            bound=roof.get("bound"), achieved=roof.get("achieved"), peak=roof.get("peak"), unit=roof.get("unit"), frac=roof.get("frac"),
            traffic=roof.get("traffic"), kernel=str(roof.get("kernel_short") or roof.get("kernel"))[:160], avg_launch_ms=roof.get("avg_launch_ms"),
            flops_per_launch=roof.get("flops_per_launch"), launches_timed=roof.get("launches_timed"),
            traffic_source=prof.get("source"), mfma_busy=dom.get("mfma_busy"), clock_ghz=dom.get("clock_ghz"),
            vendor_gemm_tflops=_get(roof, "vendor_gemm_reference", "tflops"),
            all_launches_frac=_get(roof, "all_launches", "frac"))
    else:
        line["roofline"] = None
    if cpu:
        line["cpu_baseline"] = dict(value=cpu.get("value"), unit=cpu.get("unit"), cores=cpu.get("cores"), kind=cpu.get("kind"),
                                    sample=str(cpu.get("sample"))[:260], cpu_model=cpu.get("cpu_model"), logical_cpus=cpu.get("logical_cpus"),
                                    c52_train_windows_per_s=_get(cpu, "legs", "C=52", "train_windows_per_s"))
        if cpu.get("value"):
            line["gpu_over_cpu"] = round(full.get("value", 0.0) / cpu["value"], 1)
    else:
        line["cpu_baseline"] = None
    for k in ("step_ms", "mfma_frac_whole_step", "model_tflops_per_gpu", "final_loss", "optimizer_steps_per_s", "world_size_rccl",
              "sampler_windows_per_s_per_gpu", "sampler_cosampled_windows_per_s_per_gpu"):
        if full.get(k) is not None:
            line[k] = full[k]
    # the reference's own arithmetic (fp16, train.py:98) and its own recipe (4 variables -> 52 channels, run_training.sh:39-45), same step
    for key, path in (("trainer_fp16_windows_per_s", ("module_api", "trainer_fp16", "windows_per_s")),
                      ("trainer_bf16_c52_windows_per_s", ("module_api", "trainer_bf16_c52", "windows_per_s")),
                      ("module_api_bf16_autocast_windows_per_s", ("module_api", "bf16_autocast", "windows_per_s")),
                      ("module_api_fp16_gradscaler_windows_per_s", ("module_api", "fp16_autocast_gradscaler", "windows_per_s")),
                      ("serialised_step_ms", ("by_kernel", "serialised_step_ms")),
                      ("deep_variant_train_windows_per_s", ("deep_variant", "train_windows_per_s")),
                      ("deep_variant_train_mfma_frac", ("deep_variant", "train_mfma_frac")),
                      ("sampler_member_sharded_window_forwards_per_s", ("sampler_member_sharded", "window_forwards_per_s"))):
        v = _get(full, *path)
        if v is not None:
            line[key] = v
    legs = _get(full, "sampler_configs3", "legs") or []
    short = {f"L{l['L']}_m{l['members']}_c{l['corrections']}": l.get("window_forwards_per_s") for l in legs if isinstance(l, dict) and "L" in l}
    if short:
        line["sampler_conditioned_window_forwards_per_s"] = short
    if full.get("errors"):
        line["extra_legs_failed"] = sorted(full["errors"])
    line["extras_file"] = extras_file
    text = json.dumps(line)
    if len(text) >= COMPACT_LIMIT:  # cannot happen with the fields above; if a string grew, the optional scalars go first
        for k in list(line)[::-1]:
            if k in ("extras_file",) or k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                             "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
                continue
            del line[k]
            text = json.dumps(line)
            if len(text) < COMPACT_LIMIT:
                break
    assert len(text) < COMPACT_LIMIT, len(text)
    return text


def emit(json_fd, full: dict):
    """Everything to the extras file (and nothing of it to stdout); the compact line, alone and last, to the saved stdout descriptor."""
    path = extras_path()
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        shown = os.path.relpath(path, REPO)
    except OSError as e:
        sys.stderr.write(f"bench.py: could not write {path}: {e}\n")
        shown = None
    os.write(json_fd, (compact_line(full, shown) + "\n").encode())


# ----------------------------------------------------------------------------------------------------------------- one rank
def run_rank(a):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and rank == 0:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks; reporting the ranks that run\n")
    # stdout carries exactly one line, the JSON: everything else written to file descriptor 1 -- RCCL prints its version banner
    # there from C, flushed at exit, i.e. AFTER the JSON -- goes to stderr; the JSON is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world > 1 or os.environ.get("C2W_FORCE_DIST"):  # C2W_FORCE_DIST: exercise the RCCL path with a single rank
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        world = dist.get_world_size()  # what RCCL saw is what is reported
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from climate2weather_amd import ops
    from climate2weather_amd.data import DeviceWindowFeed, SyntheticWindowDataset
    from climate2weather_amd.lr import linear_learning_rate_schedule
    from climate2weather_amd.pipelines import SDAPipeline
    from climate2weather_amd.score import ScoreUNet
    from climate2weather_amd.score_fn import BatchedScoreFunction
    from climate2weather_amd.training import Trainer

    w = 2 * a.markov_order + 1
    C = a.vars * w
    torch.manual_seed(0)
    net = ScoreUNet(channels=C, spatial=2, activation=torch.nn.SiLU, **DEFAULT_CFG).to(dev)
    total_ndata = a.batch * world * (a.steps + a.warmup + a.kernel_steps + 2) * 4
    trainer = Trainer(net, SDAPipeline(), lr_fn=lambda n: linear_learning_rate_schedule(n, total_ndata, 1e-4),
                      weight_decay=1e-3, ema_rates=[0.9999], precision=a.precision, batch_size=a.batch * world, seed=1000)
    ds = SyntheticWindowDataset(n_frames=1024 + w - 1, n_vars=a.vars, height=a.size, width=a.size, window=w, seed=0)
    feed = DeviceWindowFeed(ds, dev, rank=rank, num_replicas=world, seed=0)

    timer = LaunchTimer(ops, trainer.eng, trainer.dt, a.batch)
    timer.install()

    def one_step():
        return trainer.step(feed.next_batch(a.batch, lazy=True))  # the windows are read in place by the input conversion

    for _ in range(a.warmup):
        one_step()
    torch.cuda.synchronize()
    timer.index_layers()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.mode = "dominant"
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]  # step boundaries on the main stream (which joins the gradient stream inside every step)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        loss = one_step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.mode = "off"
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    loss_val = float(loss)
    windows = a.batch * world * a.steps
    value = windows / elapsed

    out = None
    if rank == 0:
        # Dominant kernel: the 3x3 residual-block conv at full resolution (128 -> 128 @128x128: 24 of the 70 convs of a forward and the
        # same again as input gradients).  Its forward launches have the chip to themselves; its input-gradient launches share it with
        # the weight-gradient stream (engine.grad_stream), so their durations include that interference.  The roofline is priced on the
        # launches that run alone (the kernel's own rate); the average over every launch is reported beside it (it is what
        # `rocprofv3 --stats` averages; tools/rocprof_db_stats.py splits a trace the same way).
        dom_f = timer.summarise(lambda rec, kind: kind == "fwd", steps=a.steps)
        dom_a = timer.summarise(steps=a.steps)
        gf_fwd = GFLOP_FWD.get(C, 116.0) if a.size == 128 else None
        roof = None
        if dom_f:
            (kname, kf), = list(dom_f.items())[:1]
            ka = list(dom_a.values())
            all_ms = sum(v["ms_per_step"] for v in ka)
            all_n = sum(v["launches_per_step"] for v in ka)
            all_tf = sum(v["gflop_per_launch"] * v["launches_per_step"] for v in ka) / all_ms if all_ms else 0.0
            roof = dict(bound="mfma",
                        kernel=f"conv_patch_t3_kernel<16> ({a.precision}): {kname}, residual-block conv forward launches (bias / SiLU / residual / LayerNorm "
                               "epilogues included; they run alone on the chip); network-input / output convs are NOT in this set",
                        achieved=kf["tflops"], peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=kf["frac"],
                        launches_timed=int(round(kf["launches_per_step"] * a.steps)), avg_launch_ms=kf["avg_ms"],
                        flops_per_launch=kf["gflop_per_launch"] * 1e9,
                        **pmc_traffic(a),
                        all_launches=dict(note="forward + input-gradient launches of the same layers; the latter overlap the weight-gradient stream",
                                          launches_timed=int(round(all_n * a.steps)), avg_launch_ms=round(all_ms / all_n, 4) if all_n else None,
                                          achieved=round(all_tf, 1), frac=round(all_tf / MFMA_PEAK_TFLOPS, 4)))
        out = dict(metric="UNet denoise steps/sec (train fwd+bwd+allreduce+AdamW+EMA windows/s)", value=round(value, 2), unit="windows/s",
                   n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(1e3 * elapsed / a.steps, 3), higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype=a.precision, data="synthetic",
                   config=dict(workload=f"configs/sda_unet.yml default net, {a.vars} vars x window {w} = {C} ch, {a.size}x{a.size}, "
                                        f"{a.precision} training step, {a.batch} windows/GPU/step",
                               global_batch=a.batch * world, parallelism=f"dp{world}", params=sum(p.numel() for p in net.parameters())),
                   world_size_rccl=dist.get_world_size() if dist.is_initialized() else 1,
                   optimizer_steps_per_s=round(a.steps / elapsed, 4), final_loss=round(loss_val, 5),
                   step_ms=_step_stats([marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]), roofline=roof)
        if gf_fwd:
            tf = value * (3 * gf_fwd - 1.96) / 1e3  # fwd + dgrad + wgrad minus the input conv's unused dgrad
            out["model_tflops_per_gpu"] = round(tf / world, 1)
            out["mfma_frac_whole_step"] = round(tf / world / MFMA_PEAK_TFLOPS, 4)

    extras = not a.no_extras
    # ---- what a plain dense GEMM of the vendor library reaches on THIS chip in THIS run (outside the headline region): the clock the
    # power governor holds under matrix load caps every bf16 kernel well below the 2.5 PFLOP/s spec peak `roofline.frac` is priced at
    def vendor_gemm():
        td = torch.bfloat16 if a.precision == "bf16" else torch.float16
        n = 8192
        ga, gb = torch.randn(n, n, device=dev).to(td), torch.randn(n, n, device=dev).to(td)
        for _ in range(3):
            torch.matmul(ga, gb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.matmul(ga, gb)
        e1.record()
        torch.cuda.synchronize()
        gemm_tf = 2.0 * n ** 3 / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12
        out["roofline"]["vendor_gemm_reference"] = dict(
            note=f"torch.matmul (hipBLASLt) {n}^3 {a.precision}, random operands, timed in this run after the headline region: the practical dense "
                 "matrix-pipe rate of this chip at the clock it holds; NOT the peak `frac` is priced at",
            tflops=round(gemm_tf, 1), frac_of_spec_peak=round(gemm_tf / MFMA_PEAK_TFLOPS, 4),
            dominant_kernel_vs_vendor_gemm=round(out["roofline"]["achieved"] / gemm_tf, 4))
        del ga, gb
    if extras and out is not None and out.get("roofline") and a.precision in ("bf16", "fp16"):
        _leg(out, "vendor_gemm", vendor_gemm)
    # ---- by_kernel (outside the headline region): every implicit-GEMM launch of `kernel_steps` more steps, streams serialised
    def by_kernel():
        timer.events.clear()
        prev = trainer.eng.use_grad_stream
        trainer.eng.use_grad_stream = False  # weight gradients on the caller's stream: every kernel alone on the chip
        one_step()
        torch.cuda.synchronize()
        timer.mode = "all"
        ts0 = time.perf_counter()
        for _ in range(a.kernel_steps):
            one_step()
        torch.cuda.synchronize()
        ser_ms = 1e3 * (time.perf_counter() - ts0) / a.kernel_steps
        timer.mode = "off"
        trainer.eng.use_grad_stream = prev
        if out is not None:
            fam = timer.summarise(steps=a.kernel_steps)
            gemm_ms = sum(v["ms_per_step"] for v in fam.values())
            out["by_kernel"] = dict(
                note="every implicit-GEMM launch (ops.conv / ops.conv_wgrad) of %d extra steps, HIP events per launch, backward streams serialised "
                     "(engine.use_grad_stream off) so each kernel runs alone; FLOP are algorithmic (real channel counts); weight-gradient times include the "
                     "split-K reduction launch; sorted by time per step" % a.kernel_steps,
                serialised_step_ms=round(ser_ms, 2), implicit_gemm_ms_per_step=round(gemm_ms, 2), everything_else_ms_per_step=round(ser_ms - gemm_ms, 2),
                kernels=[dict(kernel=k, **v) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms_per_step"])])
        timer.events.clear()
    if extras and a.kernel_steps > 0:
        _leg(out, "by_kernel", by_kernel)
        timer.mode = "off"

    # ---- sampler legs (outside the headline region): window-forwards/s inside the device-resident sampler
    def sampler_legs():
        net.precision = a.precision
        L = 128 + w - 1
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):  # stdout carries exactly one line: the JSON below
            sf = BatchedScoreFunction(net, markov_order=a.markov_order, batch_size=128, device=dev, noise_process=trainer.pipeline)
            sf.window_batch_floor = 0  # these legs are quoted at exactly 128 windows per network call (sampler_configs3 prices the default)
        noise = torch.randn(L, a.vars, a.size, a.size, device=dev)
        with contextlib.redirect_stdout(io.StringIO()):
            trainer.pipeline.sample(sf, noise, steps=1, show_progressbar=False)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            trainer.pipeline.sample(sf, noise, steps=a.sample_steps, show_progressbar=False)
            torch.cuda.synchronize()
        dts = time.perf_counter() - ts
        if out is not None:
            out["sampler_windows_per_s_per_gpu"] = round((L - w + 1) * a.sample_steps / dts, 1)
        # BASELINE configs[3]: 64-member ensembles, 8 members per GPU, L = 49: the members' windows share the network batches
        members, Ls = 8, 36 + w
        noise = torch.randn(members, Ls, a.vars, a.size, a.size, device=dev)
        nst = max(a.sample_steps, 6)
        with contextlib.redirect_stdout(io.StringIO()):
            trainer.pipeline.sample(sf, noise, steps=3, show_progressbar=False)  # the side streams' allocator pools fill on first use
            torch.cuda.synchronize()
            ts = time.perf_counter()
            trainer.pipeline.sample(sf, noise, steps=nst, show_progressbar=False)
            torch.cuda.synchronize()
        dts = time.perf_counter() - ts
        if out is not None:
            out["sampler_cosampled_windows_per_s_per_gpu"] = round(members * (Ls - w + 1) * nst / dts, 1)
        # north_star's second half, "ensemble sampling shards members embarrassingly" (exp/downscaling.py:96-99,248-250): every rank
        # has just sampled ITS `members` members with no collective on the data path; the job's rate is all members over the slowest rank
        if dist.is_initialized():
            dist.barrier()
            tm = torch.tensor([dts], device=dev, dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dts_job = float(tm.item())
        else:
            dts_job = dts
        if out is not None:
            out["sampler_member_sharded"] = dict(
                note="BASELINE configs[3] shape: members sharded by rank, %d co-sampled members per GPU, L = %d, %d sampler steps, no collective; "
                     "whole-job window-forwards/s = all ranks' windows over the slowest rank's time" % (members, Ls, nst),
                n_gpus=world, members_total=members * world, scaling="weak",
                window_forwards_per_s=round(world * members * (Ls - w + 1) * nst / dts_job, 1),
                members_per_hour_at_256_steps=round(world * members * 3600.0 / (256 * dts_job / nst), 1))
    if extras and a.sample_steps > 0 and a.size == 128:
        _leg(out, "sampler_legs", sampler_legs)

    # ---- BASELINE configs[4] (outside the headline region): deep variant, 80 ch x 256x256, fp16 MFMA, hipGraph-replayed sampler step
    if extras and a.size == 128 and world == 1 and not a.light_extras:
        timer.uninstall()
        del trainer, feed, ds, timer
        _empty_cache()
        import gc
        out["deep_variant"] = _leg(out, "deep_variant", lambda: deep_variant(dev))
        gc.collect()
        _empty_cache()
        out["module_api"] = _leg(out, "module_api", lambda: module_api_child(a, out["value"]))
        gc.collect()
        _empty_cache()
        out["sampler_configs3"] = _leg(out, "sampler_configs3", lambda: sampler_configs3(dev, a.precision if a.precision != "fp32" else "bf16"))

    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = _leg(out, "cpu_baseline", lambda: cpu_baseline(C, a.size, DEFAULT_CFG))
        else:
            out["cpu_baseline"] = None
        emit(json_fd, out)
    if dist.is_initialized():
        dist.destroy_process_group()


def module_api(dev, a, trainer_windows_per_s, legs=("bf16_autocast", "fp16_autocast_gradscaler", "trainer_fp16", "trainer_bf16", "trainer_bf16_c52"), item=True, lazy=False,
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

    def timed(step):
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
        return dict(windows_per_s=round(B / dt, 1), ms_per_step=round(1e3 * dt, 3), vs_trainer=round(B / dt / trainer_windows_per_s, 4),
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
    for leg, prec, nvars in (("trainer_fp16", "fp16", a.vars), ("trainer_bf16", "bf16", a.vars), ("trainer_bf16_c52", "bf16", 4)):
        if leg not in legs:
            continue
        torch.manual_seed(0)
        net = ScoreUNet(channels=nvars * w, spatial=2, activation=torch.nn.SiLU, **DEFAULT_CFG).to(dev)
        tr = Trainer(net, SDAPipeline(), lr_fn=lambda n: linear_learning_rate_schedule(n, total_ndata, 1e-4), weight_decay=1e-3, ema_rates=[0.9999],
                     precision=prec, batch_size=B, seed=1000)
        ds = SyntheticWindowDataset(n_frames=1024 + w - 1, n_vars=nvars, height=a.size, width=a.size, window=w, seed=0)
        feed = DeviceWindowFeed(ds, dev, seed=0)
        r = timed(lambda: tr.step(feed.next_batch(B, lazy=True)))
        r.update(optimizer_steps_taken=tr.optimizer_steps_taken(), loss_scale=tr.loss_scale(), channels=nvars * w)
        res[leg] = r
        del tr, net, feed, ds
    # the like-for-like ratio: every leg over the fused bf16 Trainer timed in THIS process, minutes after the headline and on the
    # same allocator state (vs_trainer compares with the headline line, taken in another process at another moment)
    if "trainer_bf16" in res:
        for k, v in res.items():
            if isinstance(v, dict) and "windows_per_s" in v:
                v["vs_trainer_same_process"] = round(v["windows_per_s"] / res["trainer_bf16"]["windows_per_s"], 4)
    return res


def module_api_child(a, trainer_windows_per_s):
    """module_api() in a process of its own (this one stays alive and idle meanwhile, its cached memory returned): the same
    conditions the headline number was taken under.  The child is this file with --module-api-child; it prints one JSON line."""
    cmd = [sys.executable, os.path.abspath(__file__), "--module-api-child", str(trainer_windows_per_s), "--steps", str(a.steps), "--warmup", str(a.warmup),
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


def main():
    a = parse()
    if a.module_api_child is not None:
        torch.cuda.set_device(0)
        print(json.dumps(module_api(torch.device("cuda", 0), a, a.module_api_child)), flush=True)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(a))
    run_rank(a)


if __name__ == "__main__":
    main()
