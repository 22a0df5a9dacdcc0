"""Throughput of the reference-style loop (only the class names change; torch autograd, torch.optim.AdamW,
StandardEMA, autocast) next to the fused Trainer, same network / batch as bench.py."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd.ema import StandardEMA
from climate2weather_amd.pipelines import SDAPipeline
from climate2weather_amd.score import ScoreUNet

dev = torch.device("cuda:0")
CFG = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3, padding_mode="zeros", attention_levels=[4])
B = int(os.environ.get("B", "128"))
torch.manual_seed(0)
net = ScoreUNet(channels=65, spatial=2, activation=torch.nn.SiLU, **CFG).to(dev)
ema = StandardEMA(net, rates=[0.9999])
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-3)
pipe = SDAPipeline()
x = torch.randn(B, 65, 128, 128, device=dev) * 0.5 + 0.5


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = pipe.loss(net, x).mean()
    loss.backward()
    opt.step()
    ema.update()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"module-API loop B={B}: {1e3 * dt:.1f} ms/step  {B / dt:.1f} windows/s  loss {l.item():.4f}")
