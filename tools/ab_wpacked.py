"""Plain [rows][9][Cin] weights against the stage-major packed copy (C2W_CONV_WPACKED) on the 16x16-tile conv kernel: interleaved timing in
one process and bit-equality of the results.   python tools/ab_wpacked.py      (ROUNDS, B; ACT=1 for bias + SiLU)"""
import math, os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd import ops

dev = torch.device("cuda:0")
B, ROUNDS, ACT = int(os.environ.get("B", "128")), int(os.environ.get("ROUNDS", "6")), int(os.environ.get("ACT", "0"))
SHAPES = [(128, 128, 128, 128), (64, 128, 128, 128), (32, 256, 256, 256), (64, 256, 128, 128), (32, 384, 256, 256), (128, 128, 128, 65)]  # H, Cin, Cout, wrows
for (H, Cin, Cout, wrows) in SHAPES:
    g = dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, Cout=Cout, ldy=Cout, wrows=wrows, mode=ops.CONV_S1)
    if not ops.conv_wpacked_supported(g, ops.DTYPE_BF16):
        print(f"H={H} {Cin}->{wrows}({Cout}): packed weights not supported for this launch")
        continue
    x = torch.randn(B * H * H, Cin, device=dev).bfloat16()
    w = (torch.randn(wrows, 9, Cin, device=dev) / math.sqrt(9 * Cin)).bfloat16()
    bias = torch.randn(wrows, device=dev)
    wp = torch.empty(ops.packed_conv_weights_numel(wrows, Cin), device=dev, dtype=torch.bfloat16)
    desc = torch.tensor([[0, 0, wrows, Cin]], dtype=torch.int64, device=dev)
    ops.pack_conv_weights_batched(w, wp, desc, 1, ops.DTYPE_BF16)
    y0, y1 = torch.empty(B * H * H, Cout, device=dev, dtype=torch.bfloat16), torch.empty(B * H * H, Cout, device=dev, dtype=torch.bfloat16)
    act = ops.ACT_SILU if ACT else ops.ACT_NONE
    runs = [lambda: ops.conv(x, w, bias, y0, g, ops.DTYPE_BF16, act=act), lambda: ops.conv(x, wp, bias, y1, g, ops.DTYPE_BF16, act=act, wpacked=True)]
    for f in runs:
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    t = [[], []]
    for r in range(ROUNDS):
        for i, f in enumerate(runs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            t[i].append(e0.elapsed_time(e1) / 10 * 1e3)
    m0, m1 = statistics.median(t[0]), statistics.median(t[1])
    print(f"H={H:3d} {Cin:3d}->{wrows:3d}({Cout}): plain {m0:7.1f} us  packed {m1:7.1f} us  ({(m1 / m0 - 1) * 100:+.1f} %)  bit-equal {torch.equal(y0, y1)}", flush=True)
