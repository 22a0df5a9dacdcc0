"""Sampler throughput at the reference's shipped trajectory lengths (exp/configs: L = 49 / 121, F = 4, k = 6, 128x128):
window-forwards/s and sampler steps/s of the device-resident predictor loop, eager launches vs hipGraph replay."""
import argparse, contextlib, io, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd.pipelines import SDAPipeline
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.score_fn import BatchedScoreFunction

p = argparse.ArgumentParser()
p.add_argument("--lengths", default="49,121")
p.add_argument("--steps", type=int, default=16)
p.add_argument("--batch", type=int, default=128)
p.add_argument("--graph", type=int, default=0)
p.add_argument("--size", type=int, default=128, help="spatial size (256: the deep variant of BASELINE configs[4])")
p.add_argument("--vars", type=int, default=4)
p.add_argument("--order", type=int, default=6, help="markov order k (window 2k + 1)")
p.add_argument("--members", type=int, default=1, help="co-sampled ensemble members (their windows share the network batches)")
p.add_argument("--precision", default="bf16", help="bf16 | fp16 (BASELINE configs[4]) | fp32")
a = p.parse_args()
dev = torch.device("cuda:0")
CFG = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3, padding_mode="zeros", attention_levels=[4])
torch.manual_seed(0)
W = 2 * a.order + 1
net = ScoreUNet(channels=a.vars * W, spatial=2, activation=torch.nn.SiLU, **CFG).to(dev).eval()
net.precision = a.precision
pipe = SDAPipeline()
for L in [int(v) for v in a.lengths.split(",")]:
    sf = BatchedScoreFunction(net, markov_order=a.order, batch_size=a.batch, device=dev, noise_process=pipe)
    if a.graph:
        sf.use_graphs = True
    noise = torch.randn(L, a.vars, a.size, a.size, device=dev) if a.members == 1 else torch.randn(a.members, L, a.vars, a.size, a.size, device=dev)
    with contextlib.redirect_stdout(io.StringIO()):
        pipe.sample(sf, noise, steps=2, show_progressbar=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.sample(sf, noise, steps=a.steps, show_progressbar=False)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nwin = (L - W + 1) * a.members
    print(f"L={L} members={a.members} windows={nwin} graph={a.graph}: {a.steps / dt:8.2f} sampler steps/s  {nwin * a.steps / dt:9.1f} window-forwards/s  {1e3 * dt / a.steps:7.2f} ms/step", flush=True)
