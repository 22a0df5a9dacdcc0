"""Host time between the per-step synchronisation of the reference's loop (training_loop.py:385: loss.item()) and the first full-chip
launch of the next step -- the part of the drop-in path's step the GPU sits idle for.  Prints per-phase host times (ms, median of N)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from climate2weather_amd import ops  # noqa: E402
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
scaler = torch.amp.GradScaler("cuda") if "--fp16" in sys.argv else None
ac = torch.float16 if scaler is not None else torch.bfloat16
feed = DeviceWindowFeed(SyntheticWindowDataset(n_frames=1024 + w - 1, n_vars=5, height=128, width=128, window=w, seed=0), dev, seed=0)
first_conv = [None]
orig_conv = ops.conv


def conv(x, wt, bias, y, g, dtype, **kw):
    if first_conv[0] is None and g["Hout"] == 128 and dtype != ops.DTYPE_F32:
        first_conv[0] = time.perf_counter()
    return orig_conv(x, wt, bias, y, g, dtype, **kw)


ops.conv = conv
rows = []
for it in range(12):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    ema.update(cur_ndata=0, batch_size=B); t.append(time.perf_counter())
    optimizer.zero_grad(); t.append(time.perf_counter())
    data = feed.next_batch(B); t.append(time.perf_counter())
    first_conv[0] = None
    with torch.autocast("cuda", dtype=ac):
        loss = pipeline.loss(net=net, x=data).mean().mul(1.0)
    t.append(time.perf_counter())
    (scaler.scale(loss) if scaler is not None else loss).backward(); t.append(time.perf_counter())
    for g in optimizer.param_groups:
        g["lr"] = 1e-4
    if scaler is not None:
        scaler.step(optimizer)
        scaler.update()
    else:
        optimizer.step()
    t.append(time.perf_counter())
    v = loss.detach().item(); t.append(time.perf_counter())
    rows.append([1e3 * (b - a) for a, b in zip(t, t[1:])] + [1e3 * (first_conv[0] - t[0])])
names = ["ema.update", "zero_grad", "next_batch", "loss (forward enqueue)", "backward enqueue", "optimizer.step", "loss.item() wait", "sync -> first 128x128 conv launch"]
rows = rows[3:]
for i, n in enumerate(names):
    col = sorted(r[i] for r in rows)
    print(f"{n:36s} median {col[len(col) // 2]:8.3f} ms   min {col[0]:8.3f}")
