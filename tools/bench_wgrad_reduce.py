import os, sys, math
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd import ops
dev = torch.device("cuda:0")
B=128
ops.ensure_workspace(dev)
for (H, Cin, Cout) in [(128,128,128),(64,128,128),(32,256,256),(16,384,384)]:
    g = dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, Cout=Cout, ldy=Cout, wrows=Cout, mode=ops.CONV_S1)
    x = torch.randn(B*H*H, Cin, device=dev).bfloat16(); y = torch.randn(B*H*H, Cout, device=dev).bfloat16()
    dw = torch.zeros(Cout*9*Cin, device=dev)
    def fn(): ops.conv_wgrad(x, y, dw, g, ops.DTYPE_BF16)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"wgrad+reduce H={H} {Cin}->{Cout}: {e0.elapsed_time(e1)/20*1e3:8.1f} us")
