#!/usr/bin/env python3
"""Headline benchmark: UNet denoise steps/sec = training windows/s through the reference's training step
(training_loop.py:369-391: noise -> ScoreUNet fwd -> MSE -> bwd -> grad all-reduce -> AdamW -> EMA) on the
default configs/sda_unet.yml network, synthetic (B, F*w, 128, 128) fields, bf16 compute, one process per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline` is measured live (HIP events on the launch stream around every launch of
the dominant kernel inside the timed region); `cpu_baseline` is the CPU oracle (oracle/, a plain-PyTorch restatement
of the same step) timed on this box's host cores, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

DEFAULT_CFG = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3,
                   padding_mode="zeros", attention_levels=[4])  # configs/sda_unet.yml
GFLOP_FWD = {65: 116.98, 52: 116.00}  # algorithmic GFLOP per (C,128,128) window forward
MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 (MI355X_MICROARCH.md)


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
    return p.parse_args()


class KernelTimer:
    """HIP events (on torch's current stream = the stream the kernels are launched on) around every launch of one
    kernel shape."""

    def __init__(self, ops, match):
        self.ops, self.match, self.events, self.enabled = ops, match, [], False
        self._orig = ops.conv

    def install(self):
        def conv(x, w, bias, y, g, dtype, **kw):
            if self.enabled and self.match(g, dtype):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._orig(x, w, bias, y, g, dtype, **kw)
                e1.record()
                self.events.append((e0, e1, bias is not None))  # forward launches carry a bias, input-gradient launches do not
            else:
                self._orig(x, w, bias, y, g, dtype, **kw)
        self.ops.conv = conv

    def mean_ms(self, forward_only=False):
        ts = [a.elapsed_time(b) for a, b, fwd in self.events if fwd or not forward_only]
        return (sum(ts) / len(ts), len(ts)) if ts else (None, 0)


def cpu_baseline(C, size, cfg):
    """The oracle (CPU port of the reference step: noise -> net -> loss -> backward) on a bounded sample."""
    from oracle import diffusion as od
    from oracle import unet as ou
    from climate2weather_amd.score import ScoreUNet
    torch.manual_seed(0)
    net = ScoreUNet(channels=C, spatial=2, activation=torch.nn.SiLU, **cfg)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    B = 2
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, C, size, size, generator=g) * 0.5 + 0.5
    fwd = lambda a, b: ou.score_unet_forward(sd, a, b, cfg["hidden_blocks"], cfg["attention_levels"])
    times = []
    for it in range(5):
        t = torch.rand(B, 1, 1, 1, generator=g)
        eps = torch.randn(B, C, size, size, generator=g)
        t0 = time.time()
        loss = od.loss(fwd, x, t, eps).mean()
        torch.autograd.grad(loss, list(sd.values()))
        times.append(time.time() - t0)
    best = sorted(times[1:])[len(times[1:]) // 2]
    return dict(value=round(B / best, 3), unit="windows/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle fwd+bwd of the same step, B={B}, C={C}, {size}x{size}, fp32, 1 warm-up + 4 timed iterations (median)")


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
    total_ndata = a.batch * world * (a.steps + a.warmup) * 4
    trainer = Trainer(net, SDAPipeline(), lr_fn=lambda n: linear_learning_rate_schedule(n, total_ndata, 1e-4),
                      weight_decay=1e-3, ema_rates=[0.9999], precision=a.precision, batch_size=a.batch * world)
    ds = SyntheticWindowDataset(n_frames=64 + w - 1, n_vars=a.vars, height=a.size, width=a.size, window=w, seed=0)
    feed = DeviceWindowFeed(ds, dev, rank=rank, num_replicas=world, seed=0)
    torch.manual_seed(1000 + rank)

    timer = KernelTimer(ops, lambda g, dt: g["mode"] == ops.CONV_S1 and g["Cin"] == 128 and g["Cout"] == 128 and g["Hin"] == a.size
                        and g["B"] == a.batch)
    timer.install()

    def one_step():
        return trainer.step(feed.next_batch(a.batch))

    for _ in range(a.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    loss_val = float(loss)
    windows = a.batch * world * a.steps
    value = windows / elapsed

    out = None
    if rank == 0:
        # Dominant kernel shape: 3x3 128->128 at full resolution.  Its forward launches have the chip to themselves; its
        # input-gradient launches share it with the weight-gradient stream (engine.grad_stream), so their durations include
        # that interference.  The roofline is priced on the launches that run alone (the kernel's own rate); the average over
        # every launch is reported beside it (it is what `rocprofv3 --stats` averages; tools/rocprof_db_stats.py splits the
        # trace the same way).
        k_ms, k_n = timer.mean_ms(forward_only=True)
        a_ms, a_n = timer.mean_ms()
        gf_fwd = GFLOP_FWD.get(C, 116.0) if a.size == 128 else None
        flops_launch = 2.0 * a.batch * a.size * a.size * 128 * 9 * 128
        roof = None
        if k_ms:
            ach = flops_launch / (k_ms * 1e-3) / 1e12
            roof = dict(bound="mfma", kernel=f"conv_patch_t3_kernel<16> ({a.precision}) 128->128 @%dx%d, res-block conv forward launches (bias / SiLU / residual / "
                        "LayerNorm epilogues included; they run alone on the chip)" % (a.size, a.size),
                        achieved=round(ach, 1), peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=round(ach / MFMA_PEAK_TFLOPS, 4),
                        # HBM bytes per launch from the PMC passes of the same kernel and shape (FETCH_SIZE x2 gfx950 correction +
                        # WRITE_SIZE; profiles/r01k_pmc_conv_patch3_b128.md) -- equal to the algorithmic 537 MB in + 537 MB out
                        traffic=1.059e9 if (a.size == 128 and a.batch == 128 and a.precision == "bf16") else None,
                        traffic_source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/r01k_pmc_conv_patch3_b128.md",
                        mfma_busy_pmc=0.61, clock_ghz_under_load_pmc=1.73,
                        launches_timed=k_n, avg_launch_ms=round(k_ms, 4), flops_per_launch=flops_launch,
                        all_launches=dict(note="forward + input-gradient launches; the latter overlap the weight-gradient stream",
                                          launches_timed=a_n, avg_launch_ms=round(a_ms, 4),
                                          achieved=round(flops_launch / (a_ms * 1e-3) / 1e12, 1),
                                          frac=round(flops_launch / (a_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)))
        out = dict(metric="UNet denoise steps/sec (train fwd+bwd+allreduce+AdamW+EMA windows/s)", value=round(value, 2), unit="windows/s",
                   n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(1e3 * elapsed / a.steps, 3), higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype=a.precision, data="synthetic",
                   config=dict(workload=f"configs/sda_unet.yml default net, {a.vars} vars x window {w} = {C} ch, {a.size}x{a.size}, "
                                        f"{a.precision} training step, {a.batch} windows/GPU/step",
                               global_batch=a.batch * world, parallelism=f"dp{world}", params=sum(p.numel() for p in net.parameters())),
                   optimizer_steps_per_s=round(a.steps / elapsed, 4), final_loss=round(loss_val, 5), roofline=roof)
        if gf_fwd:
            tf = value * (3 * gf_fwd - 1.96) / 1e3  # fwd + dgrad + wgrad minus the input conv's unused dgrad
            out["model_tflops_per_gpu"] = round(tf / world, 1)
            out["mfma_frac_whole_step"] = round(tf / world / MFMA_PEAK_TFLOPS, 4)

    # ---- sampler leg (outside the headline region): window-forwards/s inside the device-resident sampler
    if a.sample_steps > 0 and a.size == 128:
        net.precision = a.precision
        L = 128 + w - 1
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):  # stdout carries exactly one line: the JSON below
            sf = BatchedScoreFunction(net, markov_order=a.markov_order, batch_size=128, device=dev, noise_process=trainer.pipeline)
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

    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(C, a.size, DEFAULT_CFG)
        else:
            out["cpu_baseline"] = None
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
