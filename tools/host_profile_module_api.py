"""cProfile of the host side of the reference-shaped loop (what runs between the per-step synchronisation and the first launches)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from climate2weather_amd.data import DeviceWindowFeed, SyntheticWindowDataset  # noqa: E402
from climate2weather_amd.ema import StandardEMA  # noqa: E402
from climate2weather_amd.optim import AdamW  # noqa: E402
from climate2weather_amd.pipelines import SDAPipeline  # noqa: E402
from climate2weather_amd.score import ScoreUNet  # noqa: E402

dev = torch.device("cuda", 0)
B, C, w = 128, 65, 13
torch.manual_seed(0)
net = ScoreUNet(channels=C, spatial=2, activation=torch.nn.SiLU, **bench.DEFAULT_CFG).to(dev)
pipeline, optimizer, ema = SDAPipeline(), AdamW(params=net.parameters(), lr=1e-4, weight_decay=1e-3, betas=[0.9, 0.999]), StandardEMA(net=net)
feed = DeviceWindowFeed(SyntheticWindowDataset(n_frames=1024 + w - 1, n_vars=5, height=128, width=128, window=w, seed=0), dev, seed=0)


def step():
    optimizer.zero_grad()
    data = feed.next_batch(B)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = pipeline.loss(net=net, x=data).mean().mul(1.0)
    loss.backward()
    optimizer.step()
    v = loss.detach().item()
    ema.update(cur_ndata=0, batch_size=B)


for _ in range(3):
    step()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
