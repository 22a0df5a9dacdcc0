"""Upper bound for running the two halves of a batch as concurrent forward + backward passes on two HIP streams (their under-filled
small-resolution launches and their HBM-heavy epilogues would interleave): two independent trainers with B = 64 each, stepped on one
stream after the other vs on two streams at once.  No shared gradient buffer here -- this only measures what concurrency could buy."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from climate2weather_amd.pipelines import SDAPipeline
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.training import Trainer
dev = torch.device("cuda:0")
HB = int(os.environ.get("HB", "64"))
trs, xs = [], []
for i in range(2):
    torch.manual_seed(i)
    net = ScoreUNet(channels=65, spatial=2, activation=torch.nn.SiLU, **bench.DEFAULT_CFG).to(dev)
    trs.append(Trainer(net, SDAPipeline(), lr=1e-4, weight_decay=1e-3, ema_rates=[0.9999], precision="bf16", seed=1000 + i))
    xs.append(torch.randn(HB, 65, 128, 128, device=dev) * 0.5 + 0.5)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(concurrent, n=8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for i in range(2):
            if concurrent:
                with torch.cuda.stream(streams[i]):
                    trs[i].step(xs[i])
            else:
                trs[i].step(xs[i])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for tr in trs:
    tr._step_done = []
run(False, 2); run(True, 2)
for name, c in (("one stream ", False), ("two streams", True), ("one stream ", False), ("two streams", True)):
    print(f"{name}: {run(c):7.2f} ms per pair of B={HB} steps  ({2 * HB / run(c) * 1e3:7.1f} windows/s)", flush=True)
