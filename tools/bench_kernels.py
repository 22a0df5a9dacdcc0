#!/usr/bin/env python3
"""Micro-benchmarks of the HIP GEMM kernels at the default network's layer shapes (random data).
    python tools/bench_kernels.py [--batch 32]
Prints one line per (kernel, shape): ms, TFLOP/s."""
import argparse
import math
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climate2weather_amd import ops

p = argparse.ArgumentParser()
p.add_argument("--batch", type=int, default=32)
p.add_argument("--iters", type=int, default=10)
p.add_argument("--dtypes", default="bf16,fp32")
p.add_argument("--only", type=int, default=-1, help="run only SHAPES[i]")
p.add_argument("--kind", default="conv,wgrad,ln")
p.add_argument("--act", type=int, default=1)
p.add_argument("--epilogues", action="store_true", help="time the conv with each fused epilogue variant")
a = p.parse_args()
dev = torch.device("cuda:0")
WS = ops.new_workspace(dev)  # split-K scratch of the weight-gradient launches (as the engine hands it over)


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


SHAPES = [  # (mode, H, Cin, Cout)
    (ops.CONV_S1, 128, 128, 128), (ops.CONV_S1, 64, 128, 128), (ops.CONV_S1, 32, 256, 256), (ops.CONV_S1, 16, 384, 384),
    (ops.CONV_S1, 8, 512, 512), (ops.CONV_S2, 128, 128, 128), (ops.CONV_UP, 64, 128, 128), (ops.CONV_S1, 128, 64, 128),
]
for dname in a.dtypes.split(","):
    dt = {"bf16": ops.DTYPE_BF16, "fp16": ops.DTYPE_F16, "fp32": ops.DTYPE_F32}[dname]
    T = ops.TORCH_DTYPE[dt]
    B = a.batch if dt != ops.DTYPE_F32 else max(1, a.batch // 8)
    for si, (mode, H, Cin, Cout) in enumerate(SHAPES):
        if a.only >= 0 and si != a.only:
            continue
        Ho = H // 2 if mode == ops.CONV_S2 else (H * 2 if mode == ops.CONV_UP else H)
        g = dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=Ho, Wout=Ho, Cout=Cout, ldy=Cout, wrows=Cout, mode=mode)
        x = torch.randn(B * H * H, Cin, device=dev).to(T)
        w = (torch.randn(Cout, 9, Cin, device=dev) / math.sqrt(9 * Cin)).to(T)
        bias = torch.randn(Cout, device=dev)
        y = torch.empty(B * Ho * Ho, Cout, device=dev, dtype=T)
        dw = torch.zeros(Cout * 9 * Cin, device=dev)
        flops = 2.0 * B * Ho * Ho * Cout * 9 * Cin
        if "conv" in a.kind:
          ms = timeit(lambda: ops.conv(x, w, bias, y, g, dt, act=a.act), a.iters)
          print(f"conv   {dname} mode={mode} B={B} H={H} {Cin}->{Cout}: {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s", flush=True)
          if a.epilogues and ops.conv_lnbwd_supported(g, dt):
              res = torch.randn(B * Ho * Ho, Cout, device=dev).to(T)
              lnx = torch.randn(B * Ho * Ho, Cout, device=dev).to(T)
              m = torch.randn(B, Cout, device=dev)
              dm = torch.zeros(B, Cout, device=dev)
              for name, kw in [("plain", dict()), ("res", dict(res=res)), ("mul+res", dict(res=res, mul=lnx, mulmode=ops.MUL_DSILU)),
                               ("ln", dict(res=res, ln=dict(x=lnx, m=m, dm=dm, ldm=Cout, eps=1e-5, unbiased=True)))]:
                  ms = timeit(lambda: ops.conv(x, w, None, y, g, dt, **kw), a.iters)
                  print(f"   epilogue {name:8s}: {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s", flush=True)
        if "wgrad" in a.kind:
          ms = timeit(lambda: ops.conv_wgrad(x, y, dw, g, dt, workspace=WS), a.iters)
          print(f"wgrad  {dname} mode={mode} B={B} H={H} {Cin}->{Cout}: {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s", flush=True)
    if "ln" not in a.kind:
        continue
    # memory-bound kernels
    npix, C = B * 128 * 128, 128
    x = torch.randn(npix, C, device=dev).to(T)
    y = torch.empty_like(x)
    m = torch.randn(B, C, device=dev)
    ms = timeit(lambda: ops.ln_forward(x, m, y, npix, 128 * 128, C, C, 1e-5, True, dt), a.iters)
    print(f"ln_fwd {dname} {npix}x{C}: {ms:8.3f} ms  {2 * x.numel() * x.element_size() / ms / 1e6:8.1f} GB/s", flush=True)
    dx = torch.empty_like(x)
    dm = torch.zeros(B, C, device=dev)
    ms = timeit(lambda: ops.ln_backward(y, x, m, y, dx, dm, npix, 128 * 128, C, C, 1e-5, True, dt), a.iters)
    print(f"ln_bwd {dname} {npix}x{C}: {ms:8.3f} ms  {4 * x.numel() * x.element_size() / ms / 1e6:8.1f} GB/s", flush=True)
