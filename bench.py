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

from bench_legs import (DEFAULT_CFG, GFLOP_FWD, GFLOP_FWD_DEEP, MFMA_PEAK_TFLOPS, LaunchTimer, _empty_cache, _step_stats,  # noqa: E402,F401
                        deep_variant, module_api, module_api_child, sampler_configs3)


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
# (the ONLY place outside tests/ and __graft_entry__.smoke() that imports oracle/: as the thing timed beside the GPU, never on the product path)
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
                      ("trainer_bf16_b64_windows_per_s", ("module_api", "trainer_bf16_b64", "windows_per_s")),
                      ("trainer_bf16_b64_mfma_frac_whole_step", ("module_api", "trainer_bf16_b64", "mfma_frac_whole_step")),
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
    # the host enqueues ~330 launches per step one step ahead of the device; a generation-2 garbage collection in the middle of that (tens of
    # thousands of live tensor / ctypes objects) shows as one 50-ms step in twenty: collect now, not inside the timed region
    import gc
    gc.collect()
    gc.disable()
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
    gc.enable()
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
        # same again as input gradients).  Since round 5 the whole step runs on ONE stream: every launch has the chip to itself, forward
        # and input-gradient alike (with C2W_WGRAD_STREAM=1 the input-gradient launches share it with the weight-gradient stream).  The
        # roofline is priced on the FORWARD launches (bias / SiLU pair / residual / LayerNorm-emission epilogues); the average over every
        # launch of the same layers, input gradients included (LayerNorm-backward epilogues), is reported beside it -- it is what
        # `rocprofv3 --stats` averages.
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
                        kernel_short=f"conv_patch_t3_kernel<16,{a.precision}> {kname}: forward launches, fused epilogues included"[:150],
                        achieved=kf["tflops"], peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=kf["frac"],
                        launches_timed=int(round(kf["launches_per_step"] * a.steps)), avg_launch_ms=kf["avg_ms"],
                        flops_per_launch=kf["gflop_per_launch"] * 1e9,
                        **pmc_traffic(a),
                        all_launches=dict(note="forward + input-gradient launches of the same layers (one stream: each runs alone; the latter carry the LayerNorm-backward epilogues)",
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
