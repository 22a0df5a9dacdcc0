"""BASELINE.json configs[4]: the deep spatio-temporal variant -- 5 variables x 16 frames = 80 channels, 256x256 windows (473 GFLOP
forward per window).  Runs the training step and the forward on one GPU (bf16), checks bf16 against the
engine's own fp32 path on one window, and prints windows/s.  (The reference's window is 2k+1 frames, so 16 is not a valid
markov window -- SURVEY.md section 0; the network does not care: channels = 80.)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd.pipelines import SDAPipeline
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.training import Trainer

dev = torch.device("cuda:0")
CFG = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3, padding_mode="zeros", attention_levels=[4])
B = int(os.environ.get("B", "32"))
PREC = os.environ.get("PREC", "bf16")  # bf16 | fp16 (BASELINE configs[4] names fp16 MFMA)
torch.manual_seed(0)
net = ScoreUNet(channels=80, spatial=2, activation=torch.nn.SiLU, **CFG).to(dev)
x1 = torch.randn(1, 80, 256, 256, device=dev)
t1 = torch.tensor([0.4], device=dev)
with torch.no_grad():
    net.precision = "fp32"
    y32 = net(x1, t1)
    net.precision = PREC
    y16 = net(x1, t1)
rel = (y16.float() - y32).abs().max().item() / y32.abs().max().item()
print(f"{PREC} vs fp32 forward, 80 ch 256x256 (attention over 256 tokens): max rel diff {rel:.3e}")
assert rel < 3e-2
tr = Trainer(net, SDAPipeline(), lr=1e-4, precision=PREC, ema_rates=[0.9999])
x = torch.randn(B, 80, 256, 256, device=dev) * 0.5 + 0.5
for _ in range(2):
    tr.step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 4
for _ in range(n):
    loss = tr.step(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"training step B={B}: {1e3 * dt:.1f} ms  {B / dt:.1f} windows/s  ({B / dt * (3 * 473.03 - 7.9) / 1e3:.0f} model TFLOP/s)  loss {float(loss):.4f}")
with torch.no_grad():
    for _ in range(2):
        net(x, torch.rand(B, device=dev))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        net(x, torch.rand(B, device=dev))
    torch.cuda.synchronize()
dtf = (time.perf_counter() - t0) / n
print(f"forward B={B}: {1e3 * dtf:.1f} ms  {B / dtf:.1f} windows/s  ({B / dtf * 473.03 / 1e3:.0f} TFLOP/s)")
