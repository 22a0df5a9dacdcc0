"""Host time needed to ENQUEUE one training step of the default network (about 700 launches through ctypes): the first steps after a
synchronise show it (7-10 ms against a 49.5 ms step); later steps block on the HIP queue and take the GPU's pace."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from climate2weather_amd.data import DeviceWindowFeed, SyntheticWindowDataset
from climate2weather_amd.lr import linear_learning_rate_schedule
from climate2weather_amd.pipelines import SDAPipeline
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.training import Trainer
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = ScoreUNet(channels=65, spatial=2, activation=torch.nn.SiLU, **bench.DEFAULT_CFG).to(dev)
tr = Trainer(net, SDAPipeline(), lr=1e-4, weight_decay=1e-3, ema_rates=[0.9999], precision="bf16", seed=1000)
ds = SyntheticWindowDataset(n_frames=64 + 12, n_vars=5, height=128, width=128, window=13, seed=0)
feed = DeviceWindowFeed(ds, dev, rank=0, num_replicas=1, seed=0)
for _ in range(3):
    tr.step(feed.next_batch(128))
torch.cuda.synchronize()
hs = []
t0 = time.perf_counter()
for _ in range(10):
    a = time.perf_counter()
    tr._step_done.clear()  # no waiting: pure enqueue time
    tr.step(feed.next_batch(128))
    hs.append(time.perf_counter() - a)
torch.cuda.synchronize()
print("host enqueue ms per step:", [round(h * 1e3, 1) for h in hs], " wall per step", round((time.perf_counter() - t0) * 100, 2))
