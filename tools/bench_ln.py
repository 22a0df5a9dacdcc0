"""LayerNorm backward / forward at the levels where they run as separate passes (256 / 384 / 512 channels at B = 128): time and
effective HBM rate (algorithmic bytes: bwd reads dy, x, dres and writes dx; fwd reads x, writes y)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd import ops

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "128"))
dt = ops.DTYPE_BF16
T = ops.TORCH_DTYPE[dt]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for H, C in [(128, 128), (64, 128), (32, 256), (16, 384), (8, 512)]:
    HW = H * H
    npix = B * HW
    x = torch.randn(npix, C, device=dev).to(T)
    dy = torch.randn(npix, C, device=dev).to(T)
    dres = torch.randn(npix, C, device=dev).to(T)
    dx = torch.empty_like(x)
    y = torch.empty_like(x)
    m = torch.randn(B, C, device=dev)
    dm = torch.zeros(B, C, device=dev)
    nbytes = x.numel() * 2
    for name, fn, mult in [
        ("bwd +dm", lambda: ops.ln_backward(dy, x, m, dres, dx, dm, npix, HW, C, C, 1e-5, True, dt), 4),
        ("bwd    ", lambda: ops.ln_backward(dy, x, None, dres, dx, None, npix, HW, C, 0, 1e-5, True, dt), 4),
        ("fwd    ", lambda: ops.ln_forward(x, m, y, npix, HW, C, C, 1e-5, True, dt), 2),
    ]:
        ms = timeit(fn)
        print(f"ln {name} B={B} {H}x{H} C={C}: {1e3 * ms:8.1f} us  {mult * nbytes / ms / 1e6:8.0f} GB/s", flush=True)
