"""Diagnostic: where the co-resident workgroups of conv_patch_t3_kernel are, phase by phase (needs a stamp build of the library,
-DC2W_EXP=16: the kernel writes s_memrealtime (100 MHz) at its start, after the first LDS-DMA issue, behind the MFMA loop and at its
end, plus HW_ID / XCC_ID, to the buffer registered with c2w_debug_set3).  Prints, per epilogue flavour of the 128->128 @128^2 launch,
the median phase lengths and, per CU, how the time divides by the number of workgroups that are inside their MFMA loop.

    C2W_LIB=climate2weather_amd/build/alt/libc2w_stamp3.so python tools/stamp_conv3.py
"""
import sys, os, math, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from climate2weather_amd import ops, _lib

dev = torch.device("cuda:0")
B, H, C = int(os.environ.get("B", "128")), 128, 128
T = torch.bfloat16
g = dict(B=B, Hin=H, Win=H, Cin=C, Hout=H, Wout=H, Cout=C, ldy=C, wrows=C, mode=ops.CONV_S1)
x = torch.randn(B * H * H, C, device=dev).to(T)
w = (torch.randn(C, 9, C, device=dev) / math.sqrt(9 * C)).to(T)
bias = torch.randn(C, device=dev)
y = torch.empty(B * H * H, C, device=dev, dtype=T)
res = torch.randn(B * H * H, C, device=dev).to(T)
nwg = B * (H // 16) ** 2
dbg = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.c2w_debug_set3.argtypes = [ctypes.c_void_p]
assert lib.c2w_debug_set3(ctypes.c_void_p(dbg.data_ptr())) == 0
m = torch.randn(B, C, device=dev)
dm = torch.zeros(B, C, device=dev)
CASES = (("plain", {}), ("silu", dict(act=ops.ACT_SILU)), ("res", dict(res=res)), ("mul+res", dict(mul=res, res=res, mulmode=ops.MUL_DSILU)),
         ("ln", dict(res=res, ln=dict(x=res, m=m, dm=dm, ldm=C, eps=1e-5, unbiased=True))))
for name, kw in CASES:
    for _ in range(3):
        ops.conv(x, w, None if name == "ln" else bias, y, g, ops.DTYPE_BF16, **kw)
    torch.cuda.synchronize()
    d = dbg.view(nwg, 16).cpu().numpy().astype(np.int64)
    t = d[:, :4].astype(np.float64) * 0.01  # us
    t -= t[:, 0].min()
    hw, xcc = d[:, 4], d[:, 5] & 0xF
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
    total = t[:, 3].max()
    pro, loop, epi = np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]), np.median(t[:, 3] - t[:, 2])
    # per CU: sweep the loop intervals
    share = np.zeros(4)
    eshare = np.zeros(4)
    resident = np.zeros(4)
    for c in np.unique(cu):
        sel = cu == c
        for (lo, hi, acc) in ((t[sel, 1], t[sel, 2], share), (t[sel, 2], t[sel, 3], eshare), (t[sel, 0], t[sel, 3], resident)):
            ev = sorted([(v, 1) for v in lo] + [(v, -1) for v in hi])
            k, last = 0, 0.0
            for v, s in ev:
                acc[min(k, 3)] += v - last
                last = v
                k += s
            acc[min(k, 3)] += total - last
    ncu = len(np.unique(cu))
    share /= ncu * total
    eshare /= ncu * total
    resident /= ncu * total
    print(f"{name:8s} launch {total:7.1f} us  {ncu} CUs  {nwg / ncu:.1f} WG/CU   median phases: prologue {pro:5.2f}  loop {loop:6.2f}  epilogue {epi:6.2f} us")
    e = d[:, [2, 6, 7, 8, 3]].astype(np.float64) * 0.01
    print(f"           wave 0: its accumulators staged {np.median((d[:, 9] - d[:, 6]) * 0.01):5.2f} us after the gather, {np.median((d[:, 7] - d[:, 9]) * 0.01):5.2f} us before the barrier opens")
    print(f"           epilogue: waves gathered {np.median(e[:, 1] - e[:, 0]):5.2f}  staged + block 0 operands {np.median(e[:, 2] - e[:, 1]):5.2f}  "
          f"block 0 stored, block 1 operands {np.median(e[:, 3] - e[:, 2]):5.2f}  block 1 stored {np.median(e[:, 4] - e[:, 3]):5.2f} us")
    print(f"           workgroups of a CU in the MFMA loop  0/1/2/3+: " + " ".join(f"{v:5.1%}" for v in share)
          + "   in the epilogue: " + " ".join(f"{v:5.1%}" for v in eshare) + "   resident: " + " ".join(f"{v:5.1%}" for v in resident))
    # how long a workgroup takes by how its CU neighbour overlaps: loop time vs fraction of the loop the neighbour also loops
    if os.environ.get("DUMP"):
        np.save(os.path.join("gpurun_out", f"stamp3_{name}.npy"), d)
