"""How far the centre frames out of c2w_conv_center are from the full output rows (tests/test_gpu_host.py full-length fold test):
fraction of elements that differ and the largest difference in units of the bound."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from climate2weather_amd.pipelines import SDAPipeline
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.score_fn import BatchedScoreFunction
dev = torch.device("cuda:0")
L, Fv, k, H = 400, 4, 6, 128
w = 2 * k + 1
torch.manual_seed(0)
net = ScoreUNet(channels=Fv * w, spatial=2, activation=torch.nn.SiLU, **bench.DEFAULT_CFG).to(dev).eval()
for prec in ("bf16", "fp16"):
    net.precision = prec
    sf = BatchedScoreFunction(net, markov_order=k, batch_size=128, device=dev, noise_process=SDAPipeline())
    sf.window_batch_floor = 0
    x = torch.randn(L, Fv, H, H, device=dev)
    eng = net._get_engine()
    with torch.no_grad():
        for t in (0.05, 0.5, 0.95):
            eng.use_center_conv = False
            a = sf(x, torch.tensor(t)).clone()
            eng.use_center_conv = True
            b = sf(x, torch.tensor(t)).clone()
            d = (a - b).abs()
            ulp = 2.0 ** (-7 if prec == "bf16" else -10)
            bound = ulp * a.abs().clamp_min(2.0 ** -10)
            print(f"{prec} t = {t}: differing {float((d != 0).float().mean()):.5f} of elements, max d / bound {float((d / bound).max()):.3f}, scale {float(a.abs().max()):.2f}", flush=True)
