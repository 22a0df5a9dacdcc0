"""The conditioned sampler with the reference's API default exact_grad=True (src/thor/score.py:44: the likelihood score differentiates
through the network) next to exact_grad=False (what the shipped experiment configs use): sampler steps/s and window evaluations/s at
L = 121, F = 4, k = 6, 128x128, bf16, window batch 128."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from climate2weather_amd.pipelines import SDAPipeline
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.score_fn import BatchedScoreFunction, PoolStrideOperator

dev = torch.device("cuda:0")
F, k, H, L = 4, 6, 128, int(os.environ.get("L", "121"))
w = 2 * k + 1
torch.manual_seed(0)
net = ScoreUNet(channels=F * w, spatial=2, activation=torch.nn.SiLU, **bench.DEFAULT_CFG).to(dev).eval()
net.precision = "bf16"
if os.environ.get("FROZEN", "1") == "1":
    net.requires_grad_(False)  # the sampler's copy in the reference: a pickled snapshot saved with requires_grad_(False) (training_loop.py:253-265)
pipe = SDAPipeline()
A = PoolStrideOperator(16, 6)
std = torch.tensor([0.1692666615037876, 0.0425178630338289, 0.3268027589410125, 0.3268027589410125]).view(1, F, 1, 1)
g = torch.Generator(device=dev).manual_seed(L)
truth = torch.randn((L, F, H, H), device=dev, generator=g) * 0.5 + 0.5
noise = torch.randn((L, F, H, H), device=dev, generator=g)
for exact in (False, True):
    with contextlib.redirect_stdout(io.StringIO()):
        sf = BatchedScoreFunction(net, markov_order=k, batch_size=128, device=dev, noise_process=pipe)
        sf.condition_on(A=A, y=A(truth), std=std, gamma=0.0007196856730011522, exact_grad=exact)
        pipe.sample(sf, noise, steps=1, corrections=0, device=dev, show_progressbar=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 4
        x = pipe.sample(sf, noise, steps=n, corrections=0, device=dev, show_progressbar=False)
        torch.cuda.synchronize()
    d = (time.perf_counter() - t0) / n
    print(f"L = {L} frozen = {os.environ.get('FROZEN', '1')} exact_grad = {exact}: {1e3 * d:8.2f} ms per sampler step, {(L - w + 1) / d:9.1f} window evaluations/s, finite {bool(torch.isfinite(x).all())}", flush=True)
