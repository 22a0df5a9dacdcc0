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
ntile = B*(H//16)**2 * 2  # 8x16-pixel tiles (lab build of conv_patch_lab.hip with -DC2W_EXP=16)
NS = 8
dbg = torch.zeros(ntile*NS, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.c2w_debug_set.argtypes = [ctypes.c_void_p]
assert lib.c2w_debug_set(ctypes.c_void_p(dbg.data_ptr())) == 0
m = torch.randn(B, C, device=dev); dm = torch.zeros(B, C, device=dev)
for name, kw in (("plain", {}), ("silu", dict(act=ops.ACT_SILU)), ("res", dict(res=res)), ("mul+res", dict(mul=res, res=res, mulmode=ops.MUL_DSILU)),
                 ("y2", dict(y2=torch.empty_like(y))), ("ln", dict(res=res, ln=dict(x=res, m=m, dm=dm, ldm=C, eps=1e-5, unbiased=True)))):
    for _ in range(3):
        ops.conv(x, w, None if name == "ln" else bias, y, g, ops.DTYPE_BF16, **kw)
    torch.cuda.synchronize()
    d = dbg.view(ntile, NS).cpu().double()
    seg = [(d[:, i+1]-d[:, i]).median().item() for i in range(NS - 1)]
    names = ["prologue", "loop", "prefetch", "barrier1", "acc->LDS", "barrier2", "store"] if NS == 8 else ["prologue", "loop", "acc->LDS", "store"]
    print(f"{name:8s} cycles/tile: " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, seg)) + f"  total {(d[:,NS-1]-d[:,0]).median().item():.0f}")
