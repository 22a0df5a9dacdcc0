"""The reference-shaped loop (training_loop.py:369-391, five class-name strings) trains: full-size network, B = 64, N steps on the
synthetic feed, bf16 autocast and fp16 autocast + torch.amp.GradScaler, beside the fused Trainer on the same feed; prints the mean
loss every 25 steps, the steps the optimizer actually took and the final loss scale."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd.data import DeviceWindowFeed, SyntheticWindowDataset
from climate2weather_amd.ema import StandardEMA
from climate2weather_amd.optim import AdamW
from climate2weather_amd.pipelines import SDAPipeline
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.training import Trainer

dev = torch.device("cuda:0")
CFG = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3, padding_mode="zeros", attention_levels=[4])
N, B = int(os.environ.get("STEPS", "150")), 64


def feed():
    return DeviceWindowFeed(SyntheticWindowDataset(n_frames=76, n_vars=5, height=128, width=128, window=13, seed=0), dev, seed=0)


for name, ac, use_scaler in (("module bf16 autocast", torch.bfloat16, False), ("module fp16 autocast + GradScaler", torch.float16, True)):
    torch.manual_seed(0)
    net = ScoreUNet(channels=65, spatial=2, activation=torch.nn.SiLU, **CFG).to(dev)
    pipeline, optimizer, ema = SDAPipeline(), AdamW(params=net.parameters(), lr=2e-4, weight_decay=1e-3, betas=[0.9, 0.999]), StandardEMA(net=net)
    scaler = torch.amp.GradScaler("cuda", growth_interval=50) if use_scaler else None
    f = feed()
    torch.manual_seed(1)
    losses = []
    for s in range(N):
        optimizer.zero_grad()
        with torch.autocast("cuda", dtype=ac):
            loss = pipeline.loss(net=net, x=f.next_batch(B)).mean()
        (scaler.scale(loss) if scaler is not None else loss).backward()
        if scaler is not None:
            scaler.step(optimizer)
            scaler.update()
        else:
            optimizer.step()
        losses.append(loss.detach())
        ema.update()
        if (s + 1) % 25 == 0:
            print(f"{name} step {s + 1:4d}: loss {float(torch.stack(losses[-25:]).mean()):.4f}", flush=True)
    fin = all(bool(torch.isfinite(p).all()) for p in net.parameters())
    print(f"{name}: optimizer steps taken {optimizer.steps_taken()} of {N}, loss scale {scaler.get_scale() if scaler else 1:g}, finite {fin}, "
          f"flat path {optimizer.fused_path_active()}")
    del net, optimizer, ema, f
for prec in ("bf16",):
    torch.manual_seed(0)
    net = ScoreUNet(channels=65, spatial=2, activation=torch.nn.SiLU, **CFG).to(dev)
    tr = Trainer(net, SDAPipeline(), lr=2e-4, precision=prec, ema_rates=[0.9999])
    f = feed()
    torch.manual_seed(1)
    losses = []
    for s in range(N):
        losses.append(tr.step(f.next_batch(B)))
        if (s + 1) % 25 == 0:
            print(f"Trainer {prec} step {s + 1:4d}: loss {float(torch.stack(losses[-25:]).mean()):.4f}", flush=True)
