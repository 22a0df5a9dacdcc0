"""Short trajectories (exp/configs: L = 49 / 121 -> 37 / 109 windows per score evaluation): one member's windows as ONE network call
against the same windows split into 2 ... 4 equal batches on concurrent HIP streams (the launches of the 8x8 / 16x16 levels carry
fewer workgroups than the chip has CUs and cost their K chain whatever they carry: two of them side by side cost one).

    python tools/bench_sampler_split.py [--lengths 49,121] [--splits 1,2,3,4]
"""
import argparse, contextlib, io, json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from climate2weather_amd.pipelines import SDAPipeline
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.score_fn import BatchedScoreFunction, PoolStrideOperator

p = argparse.ArgumentParser()
p.add_argument("--lengths", default="49,121")
p.add_argument("--splits", default="1,2,3,4")
p.add_argument("--precision", default="bf16")
p.add_argument("--rounds", type=int, default=3)
a = p.parse_args()
dev = torch.device("cuda:0")
F, k, H = 4, 6, 128
w = 2 * k + 1
torch.manual_seed(0)
net = ScoreUNet(channels=F * w, spatial=2, activation=torch.nn.SiLU, **bench.DEFAULT_CFG).to(dev).eval()
net.precision = a.precision
pipe = SDAPipeline()
A = PoolStrideOperator(16, 6)
std = torch.tensor([0.1692666615037876, 0.0425178630338289, 0.3268027589410125, 0.3268027589410125]).view(1, F, 1, 1)
for L in (int(v) for v in a.lengths.split(",")):
    nwin = L - w + 1
    g = torch.Generator(device=dev).manual_seed(L)
    truth = torch.randn((L, F, H, H), device=dev, generator=g) * 0.5 + 0.5
    noise = torch.randn((L, F, H, H), device=dev, generator=g)
    sfs = {}
    for ns in (int(v) for v in a.splits.split(",")):
        with contextlib.redirect_stdout(io.StringIO()):
            sf = BatchedScoreFunction(net, markov_order=k, batch_size=-(-nwin // ns), device=dev, noise_process=pipe)
            sf.window_batch_floor = 0
            sf.num_streams = max(ns, 1)
            sf.condition_on(A=A, y=A(truth), std=std, gamma=0.0007196856730011522, exact_grad=False)
            pipe.sample(sf, noise, steps=2, device=dev, show_progressbar=False)
        sfs[ns] = sf
    torch.cuda.synchronize()
    res = {ns: [] for ns in sfs}
    for _ in range(a.rounds):  # interleaved rounds in one process
        for ns, sf in sfs.items():
            with contextlib.redirect_stdout(io.StringIO()):
                t0 = time.perf_counter()
                pipe.sample(sf, noise, steps=24, device=dev, show_progressbar=False)
                torch.cuda.synchronize()
            res[ns].append((time.perf_counter() - t0) / 24)
    for ns, ts in res.items():
        d = sorted(ts)[len(ts) // 2]
        print(json.dumps(dict(L=L, windows=nwin, batches=ns, batch_size=-(-nwin // ns), ms_per_step=round(1e3 * d, 3), ms_min=round(1e3 * min(ts), 3),
                              window_forwards_per_s=round(nwin / d, 1))), flush=True)
