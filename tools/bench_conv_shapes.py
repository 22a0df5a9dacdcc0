"""Forward / input-gradient implicit-GEMM launches at every conv shape of the default network (B = 128, bf16): time, TFLOP/s and
the number of workgroup rounds, to spot shapes the dispatcher serves badly."""
import os, sys, math
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd import ops
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "128"))
dt = ops.DTYPE_BF16
S1, S2, UP, TS2, X1 = ops.CONV_S1, ops.CONV_S2, ops.CONV_UP, ops.CONV_TS2, ops.CONV_1X1
SHAPES = [  # (label, mode, Hin, Cin, Cout)
    ("head 65->128", S1, 128, 128, 128), ("res 128 @128", S1, 128, 128, 128), ("res 128 @64", S1, 64, 128, 128), ("res 256 @32", S1, 32, 256, 256),
    ("res 384 @16", S1, 16, 384, 384), ("res 512 @8", S1, 8, 512, 512),
    ("down 128->128", S2, 128, 128, 128), ("down 128->256", S2, 64, 128, 256), ("down 256->384", S2, 32, 256, 384), ("down 384->512", S2, 16, 384, 512),
    ("up 512->384 @16", S1, 16, 512, 384), ("up 384->256 @32", S1, 32, 384, 256), ("up 256->128 @64", S1, 64, 256, 128), ("up 128->128 @128", S1, 128, 128, 128),
    ("dgrad up 384->512 @16", S1, 16, 384, 512), ("dgrad up 256->384 @32", S1, 32, 256, 384), ("dgrad up 128->256 @64", S1, 64, 128, 256),
    ("dgrad down 128->128", TS2, 64, 128, 128), ("dgrad down 256->128", TS2, 32, 256, 128), ("dgrad down 384->256", TS2, 16, 384, 256), ("dgrad down 512->384", TS2, 8, 512, 384),
    ("qkv 512->1536", X1, 8, 512, 1536), ("proj 512->512", X1, 8, 512, 512), ("out 128->65(128)", S1, 128, 128, 128),
]
for label, mode, H, Cin, Cout in SHAPES:
    Ho = H // 2 if mode == S2 else (H * 2 if mode in (UP, TS2) else H)
    taps = 1 if mode == X1 else 9
    g = dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=Ho, Wout=Ho, Cout=Cout, ldy=Cout, wrows=Cout, mode=mode)
    x = torch.randn(B * H * H, Cin, device=dev).bfloat16()
    w = (torch.randn(Cout, taps, Cin, device=dev) / math.sqrt(taps * Cin)).bfloat16()
    y = torch.empty(B * Ho * Ho, Cout, device=dev, dtype=torch.bfloat16)
    def fn(): ops.conv(x, w, None, y, g, dt)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mac_pix = B * Ho * Ho if mode != TS2 else B * H * H  # TS2: every input (dy) pixel meets 9 taps
    gf = 2.0 * mac_pix * Cout * taps * Cin / 1e9
    print(f"{label:24s} mode={mode} H={H:3d} {Cin:4d}->{Cout:4d}: {us:8.1f} us  {gf / us * 1e3:7.1f} TFLOP/s", flush=True)
