"""Full-size network, fp16 training under the device-resident loss scale: N optimizer steps on the synthetic feed; prints the loss
every 25 steps, the steps actually taken (skipped ones are overflow back-offs) and the final scale; bf16 beside it."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd.data import DeviceWindowFeed, SyntheticWindowDataset
from climate2weather_amd.pipelines import SDAPipeline
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.training import Trainer

dev = torch.device("cuda:0")
CFG = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3, padding_mode="zeros", attention_levels=[4])
N = int(os.environ.get("STEPS", "150"))
for prec in ("bf16", "fp16"):
    torch.manual_seed(0)
    net = ScoreUNet(channels=65, spatial=2, activation=torch.nn.SiLU, **CFG).to(dev)
    tr = Trainer(net, SDAPipeline(), lr=2e-4, precision=prec, ema_rates=[0.9999], growth_interval=50)
    feed = DeviceWindowFeed(SyntheticWindowDataset(n_frames=76, n_vars=5, height=128, width=128, window=13, seed=0), dev, seed=0)
    torch.manual_seed(1)
    losses = []
    for s in range(N):
        losses.append(tr.step(feed.next_batch(64)))
        if (s + 1) % 25 == 0:
            print(f"{prec} step {s + 1:4d}: loss {float(torch.stack(losses[-25:]).mean()):.4f}", flush=True)
    print(f"{prec}: optimizer steps taken {tr.optimizer_steps_taken()} of {N}, loss scale {tr.loss_scale():g}, finite {bool(torch.isfinite(tr.eng.flat).all())}")
