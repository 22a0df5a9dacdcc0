"""Weight-gradient launches (kernel + split-K reduction) at the shapes of the default network, B = 128, bf16: time and TFLOP/s."""
import os, sys, math
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd import ops
dev = torch.device("cuda:0")
B=128
ws = ops.new_workspace(dev)
SHAPES = [(ops.CONV_S1, 128, 128, 128), (ops.CONV_S1, 64, 128, 128), (ops.CONV_S1, 32, 256, 256), (ops.CONV_S1, 16, 384, 384), (ops.CONV_S1, 16, 512, 384),
          (ops.CONV_S1, 32, 384, 256), (ops.CONV_S1, 8, 512, 512), (ops.CONV_S2, 128, 128, 128), (ops.CONV_S2, 64, 128, 256), (ops.CONV_S2, 32, 256, 384),
          (ops.CONV_S2, 16, 384, 512), (ops.CONV_1X1, 8, 512, 1536), (ops.CONV_1X1, 8, 512, 512)]
for (mode, H, Cin, Cout) in SHAPES:
    Ho = H // 2 if mode == ops.CONV_S2 else H
    g = dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=Ho, Wout=Ho, Cout=Cout, ldy=Cout, wrows=Cout, mode=mode)
    x = torch.randn(B*H*H, Cin, device=dev).bfloat16(); y = torch.randn(B*Ho*Ho, Cout, device=dev).bfloat16()
    taps = 1 if mode == ops.CONV_1X1 else 9
    dw = torch.zeros(Cout*taps*Cin, device=dev)
    gf = 2.0 * B * Ho * Ho * Cout * taps * Cin / 1e9
    def fn(): ops.conv_wgrad(x, y, dw, g, ops.DTYPE_BF16, workspace=ws)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"wgrad+reduce mode={mode} H={H} {Cin}->{Cout}: {us:8.1f} us  {gf / us * 1e3:7.1f} TFLOP/s")
