#!/usr/bin/env python3
"""Is the stride-2 forward kernel's OUTPUT ever wrong inside a training step?  (profiles/r06_experiments.md section 10d: launch by launch it is
bit-reproducible, yet small bf16 trainings with it show loss jumps.)  Runs the bf16 leg of
tests/test_gpu_host.py::test_bf16_and_fp16_training_track_fp32_training R times with C2W_CONV_S2_PATCH=1 and, behind every stride-2 forward
launch, the gather kernel on the same operands (naive = 2) into a second buffer: per run the largest |difference| / max|reference| over its 60
launches, the steps whose loss jumps, and -- for a launch that is off by more than 3 % of the scale -- the operands saved to
gpurun_out/s2_mismatch_<run>_<step>.pt (x, w, bias, y, y_ref).  No synchronisation inside a run (the check must not change the timing it is about
more than a second launch does).        python tools/probe_s2_forward_in_training.py [R]"""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["C2W_CONV_S2_PATCH"] = "1"
import torch
from climate2weather_amd import ops
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.training import Trainer

R = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = dict(embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3, padding_mode="zeros")
real_conv = ops.conv
calls = []  # per launch of the current run: (ratio tensor, x, w clone, bias, y, y_ref)


def checked_conv(x, w, bias, y, g, dtype, *a, **kw):
    real_conv(x, w, bias, y, g, dtype, *a, **kw)
    if g["mode"] == ops.CONV_S2 and not kw.get("naive") and ops.conv_dispatch(g, dtype) == 5:
        y_ref = torch.empty_like(y)
        kw2 = dict(kw, naive=2)
        real_conv(x, w, bias, y_ref, g, dtype, *a, **kw2)
        ratio = (y.float() - y_ref.float()).abs().max() / y_ref.float().abs().max().clamp_min(1e-20)
        calls.append((ratio, x, w.clone(), bias.clone() if bias is not None else None, y, y_ref))


ops.conv = checked_conv
os.makedirs("gpurun_out", exist_ok=True)
worst_all, jumps, saved, ncalls = 0.0, 0, 0, 0
for run in range(R):
    calls.clear()
    torch.manual_seed(11)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **cfg).cuda()
    tr = Trainer(net, lr=2e-3, precision="bf16", ema_rates=[0.999])
    gen = torch.Generator().manual_seed(3)
    base = torch.randn(8, 6, 32, 32, generator=gen) * 0.5 + 0.5
    losses = []
    for s in range(60):
        x = (base + 0.05 * torch.randn(8, 6, 32, 32, generator=gen)).cuda()
        t = torch.rand(8, generator=gen).cuda()
        eps = torch.randn(8, 6, 32, 32, generator=gen).cuda()
        losses.append(tr.step(x, t=t, eps=eps))
    torch.cuda.synchronize()
    losses = [float(v) for v in losses]
    ratios = [float(c[0]) for c in calls]
    ncalls += len(ratios)
    worst = max(ratios) if ratios else float("nan")
    worst_all = max(worst_all, worst)
    jump_steps = [i for i in range(20, 60) if losses[i] > 1.2]
    jumps += bool(jump_steps)
    off = [i for i, r in enumerate(ratios) if not (r <= 0.03)]
    if jump_steps or off:
        print(f"run {run}: {len(ratios)} stride-2 forward launches, largest difference {worst:.2e} of the scale; loss jumps at steps {jump_steps}; launches off by > 3 %: {off}; "
              f"ratios around them: {[f'{ratios[j]:.1e}' for i in (off or jump_steps)[:2] for j in range(max(0, i - 2), min(len(ratios), i + 3))]}", flush=True)
    for i in off[:2]:
        if saved < 4:
            _, x_, w_, b_, y_, yr_ = calls[i]
            torch.save(dict(x=x_.cpu(), w=w_.cpu(), bias=None if b_ is None else b_.cpu(), y=y_.cpu(), y_ref=yr_.cpu(), step=i, run=run), f"gpurun_out/s2_mismatch_{run}_{i}.pt")
            saved += 1
print(f"{R} runs, {ncalls} checked launches: largest difference between the two kernels {worst_all:.2e} of the scale; runs with a loss jump: {jumps}; operand sets saved: {saved}")
