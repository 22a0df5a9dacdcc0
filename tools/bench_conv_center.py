"""The sampler's output stage at the benchmarked size: full output convolution (52 of 128 rows) + window_scatter, against
c2w_conv_center (the centre frame's 4 rows, written into the eps planes).  B windows of 128x128, bf16."""
import math, os, sys
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd import ops
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "128"))
H = W = 128
Cin, F, k = 128, 4, 6
w_ = 2 * k + 1
wrows = F * w_
dt = ops.DTYPE_BF16
x = torch.randn(B * H * W, Cin, device=dev).bfloat16()
wt = (torch.randn(wrows, 9, Cin, device=dev) / math.sqrt(9 * Cin)).bfloat16()
bias = torch.randn(wrows, device=dev)
y = torch.zeros(B * H * W, 128, device=dev, dtype=torch.bfloat16)
eps = torch.zeros(B + 2 * k, F, H, W, device=dev)
g = dict(B=B, Hin=H, Win=W, Cin=Cin, Hout=H, Wout=W, Cout=128, ldy=128, wrows=wrows, mode=ops.CONV_S1)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def full():
    ops.conv(x, wt, bias, y, g, dt)
    ops.window_scatter(y, eps, B, F, H * W, k, 1, B + 2, 128, dt)  # interior windows


def center():
    ops.conv_center(x, wt, bias, eps[k + 1:], B, H, W, Cin, wrows, k * F, F, F * H * W, dt)


full()
a = eps.clone()
eps.zero_()
center()
d = (eps - a).abs().max().item()
print(f"B = {B}: full conv + scatter {timed(full):8.1f} us    conv_center {timed(center):8.1f} us    max |difference| {d:.3e} (scale {a.abs().max().item():.2f})")
