"""tests/test_gpu_host.py::test_bf16_and_fp16_training_track_fp32_training, N times in one process: the curve statistics the test
compares, to see how close to its bounds they run (a failure in 1 of ~10 full-suite runs was seen once)."""
import os, sys
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.training import Trainer
N = int(os.environ.get("N", "12"))
cfg = dict(embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3, padding_mode="zeros")
junk = [torch.cuda.Stream() for _ in range(int(os.environ.get("JUNK_STREAMS", "0")))]
for it in range(N):
    curves, extra = {}, {}
    for prec in ("fp32", "bf16", "fp16"):
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
        curves[prec] = losses
        if prec == "fp16":
            extra = dict(steps=tr.optimizer_steps_taken(), scale=tr.loss_scale())
    first = sum(curves["fp32"][:5]) / 5
    a = torch.tensor(curves["fp32"][-20:]).mean().item()
    line = f"run {it}: first {first:.4f}"
    for prec in curves:
        last = sum(curves[prec][-10:]) / 10
        b = torch.tensor(curves[prec][-20:]).mean().item()
        line += f" | {prec}: last/first {last / first:.3f} tail {b:.4f} ({100 * (b - a) / a:+.2f} %)"
    print(line, extra, flush=True)
