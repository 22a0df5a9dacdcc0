#!/usr/bin/env python3
"""tests/test_gpu_host.py::test_bf16_and_fp16_training_track_fp32_training, the bf16 leg R times in one process: the spread of the curve's tail
(mean of the last 20 of 60 steps) relative to the fp32 tail, and for the runs furthest off the whole curve -- a one-step jump (a corrupted
update) or a curve that runs beside the others?      python tools/probe_training_curve_spread.py [R]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.training import Trainer

R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = dict(embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3, padding_mode="zeros")


def curve(prec):
    torch.manual_seed(11)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **cfg).cuda()
    tr = Trainer(net, lr=2e-3, precision=prec, ema_rates=[0.999])
    gen = torch.Generator().manual_seed(3)
    base = torch.randn(8, 6, 32, 32, generator=gen) * 0.5 + 0.5
    losses = []
    for s in range(60):
        x = (base + 0.05 * torch.randn(8, 6, 32, 32, generator=gen)).cuda()
        t = torch.rand(8, generator=gen).cuda()
        eps = torch.randn(8, 6, 32, 32, generator=gen).cuda()
        losses.append(float(tr.step(x, t=t, eps=eps)))
    return losses


ref = curve("fp32")
a = sum(ref[-20:]) / 20
runs = [curve("bf16") for _ in range(R)]
tails = [sum(c[-20:]) / 20 for c in runs]
rel = sorted((t - a) / a for t in tails)
print(f"fp32 tail {a:.4f}; bf16 tail relative to it over {R} runs: min {rel[0]:+.3f} median {rel[len(rel) // 2]:+.3f} max {rel[-1]:+.3f}; beyond +-5 %: {sum(abs(r) > 0.05 for r in rel)}, beyond +-10 %: {sum(abs(r) > 0.10 for r in rel)}")
print("distinct bf16 curves:", len({tuple(c) for c in runs}), "; runs with a spike (a loss above 1.2 after step 20; fp32 max there: %.3f):" % max(ref[20:]), sum(max(c[20:]) > 1.2 for c in runs))
worst = max(range(R), key=lambda i: abs(tails[i] - a))
med = sorted(range(R), key=lambda i: tails[i])[R // 2]
for tag, i in (("furthest", worst), ("median", med)):
    print(f"{tag} run: tail {tails[i]:.4f}; steps 0-59:", " ".join(f"{v:.3f}" for v in runs[i]))
print("fp32:", " ".join(f"{v:.3f}" for v in ref))
