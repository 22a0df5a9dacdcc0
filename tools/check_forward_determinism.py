"""Is the inference forward run-to-run bit-identical?  Default network, bf16, B = 128 windows of C = 52 (the sampler's shape) and
C = 65; N repeats on one stream, then the L-frame score evaluation (window batches on several streams) twice.  Prints where two runs
differ (count, first indices, magnitude) -- a forward pass has no atomics, so ANY difference is a race."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd.pipelines import SDAPipeline
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.score_fn import BatchedScoreFunction

dev = torch.device("cuda:0")
CFG = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3, padding_mode="zeros", attention_levels=[4])
N = int(os.environ.get("N", "8"))
L = int(os.environ.get("L", "1200"))


def diff(a, b, what):
    ne = (a != b)
    n = int(ne.sum())
    if n:
        idx = ne.nonzero()[:5].tolist()
        d = (a.float() - b.float()).abs()
        print(f"  {what}: {n} of {a.numel()} elements differ; max |d| {d.max().item():.3e}; nan {int(torch.isnan(a).sum())}/{int(torch.isnan(b).sum())}; first {idx}", flush=True)
    return n


for C in (52, 65):
    torch.manual_seed(0)
    net = ScoreUNet(channels=C, spatial=2, activation=torch.nn.SiLU, **CFG).to(dev).eval()
    net.precision = os.environ.get("PREC", "bf16")
    x = torch.randn(128, C, 128, 128, device=dev)
    t = torch.tensor(0.6, device=dev)
    with torch.no_grad():
        y0 = net(x, t).clone()
        bad = 0
        for i in range(N):
            junk = torch.full((1 << 26,), float("nan"), device=dev)  # poison freed memory: a read of uninitialised scratch shows up
            del junk
            bad += diff(net(x, t), y0, f"C={C} repeat {i}")
    print(f"C={C}: forward B=128 {'IDENTICAL' if bad == 0 else 'DIFFERS'} over {N} repeats", flush=True)
    if C == 52:
        sf = BatchedScoreFunction(net, markov_order=6, batch_size=128, device=dev, noise_process=SDAPipeline())
        xs = torch.randn(L, 4, 128, 128, device=dev)
        with torch.no_grad():
            e0 = sf(xs, torch.tensor(0.7)).clone()
            bad = 0
            for i in range(3):
                junk = torch.full((1 << 26,), float("nan"), device=dev)
                del junk
                bad += diff(sf(xs, torch.tensor(0.7)), e0, f"score evaluation L={L} repeat {i}")
        print(f"score evaluation (multi-stream) {'IDENTICAL' if bad == 0 else 'DIFFERS'}", flush=True)
        for ns in (1,):
            sf.num_streams = ns
            with torch.no_grad():
                e1 = sf(xs, torch.tensor(0.7)).clone()
                bad = diff(e1, e0, "one stream vs several")
                for i in range(3):
                    bad += diff(sf(xs, torch.tensor(0.7)), e1, f"one-stream repeat {i}")
            print(f"score evaluation (one stream) {'IDENTICAL' if bad == 0 else 'DIFFERS'}", flush=True)
    del net
