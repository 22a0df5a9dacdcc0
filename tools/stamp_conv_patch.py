"""Diagnostic: per-tile cycle stamps of conv_patch_s1_kernel (needs a stamp build of the library: the kernel writes
s_memtime at 5 points to the buffer registered with c2w_debug_set; see profiles/r01_stamps_conv_patch.md)."""
import sys, os, math, ctypes
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd import ops, _lib
dev = torch.device("cuda:0")
B, H, C = 32, 128, 128
T = torch.bfloat16
g = dict(B=B, Hin=H, Win=H, Cin=C, Hout=H, Wout=H, Cout=C, ldy=C, wrows=C, mode=ops.CONV_S1)
x = torch.randn(B*H*H, C, device=dev).to(T); w = (torch.randn(C, 9, C, device=dev)/math.sqrt(9*C)).to(T)
bias = torch.randn(C, device=dev); y = torch.empty(B*H*H, C, device=dev, dtype=T)
res = torch.randn(B*H*H, C, device=dev).to(T)
ntile = B*(H//16)**2 * (1 if os.environ.get('C2W_CONV_FULL') else 2)
dbg = torch.zeros(ntile*5, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.c2w_debug_set.argtypes = [ctypes.c_void_p]
assert lib.c2w_debug_set(ctypes.c_void_p(dbg.data_ptr())) == 0
for name, kw in (("plain", {}), ("res", dict(res=res)), ("mul", dict(mul=res, mulmode=ops.MUL_DSILU)), ("y2", dict(y2=torch.empty_like(y)))):
    for _ in range(3):
        ops.conv(x, w, bias, y, g, ops.DTYPE_BF16, **kw)
    torch.cuda.synchronize()
    d = dbg.view(ntile, 5).cpu().double()
    seg = [(d[:, i+1]-d[:, i]).median().item() for i in range(4)]
    print(f"{name:6s} cycles/tile: prologue {seg[0]:.0f}  loop {seg[1]:.0f}  acc->LDS {seg[2]:.0f}  store {seg[3]:.0f}  total {(d[:,4]-d[:,0]).median().item():.0f}")
