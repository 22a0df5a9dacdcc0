"""A/B of two (or more) builds of the library on the weight-gradient launches of the default network (B = 128, bf16), interleaved
rounds in ONE process on one device (cdna_hip_programming.md 5.4 rule 24): median and min of the per-round times, kernel + split-K
reduction, and the difference of the results.

    python tools/ab_wgrad.py climate2weather_amd/build/libc2w_old.so climate2weather_amd/libc2w_hip.so
"""
import ctypes, os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd import _lib
from climate2weather_amd._lib import ConvArgs, c_int, c_longlong

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "128"))
ROUNDS = int(os.environ.get("ROUNDS", "7"))
S1, S2, X1 = _lib.CONV_S1, _lib.CONV_S2, _lib.CONV_1X1


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, argtypes in _lib._PROTOS.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes = argtypes
            fn.restype = c_longlong if name.endswith("_bytes") else c_int
    return lib


libs = [(os.path.basename(p), load(p)) for p in sys.argv[1:]]
SHAPES = [(S1, 128, 128, 128), (S1, 64, 128, 128), (S1, 32, 256, 256), (S1, 16, 384, 384), (S1, 8, 512, 512), (S1, 64, 256, 128), (S1, 32, 384, 256),
          (S1, 16, 512, 384), (S2, 128, 128, 128), (S2, 64, 128, 256), (S2, 32, 256, 384), (S2, 16, 384, 512), (X1, 8, 512, 1536), (X1, 8, 512, 512)]
if os.environ.get("SHAPE"):  # one custom geometry: SHAPE=mode,H,Cin,Cout (mode: S1 = 1, S2 = 2, 1x1 = 0)
    SHAPES = [tuple(int(v) for v in os.environ["SHAPE"].split(","))]
elif os.environ.get("SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["SHAPES"].split(",")]
ws = torch.empty((96 << 20) // 4, dtype=torch.float32, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (mode, H, Cin, Cout, *rest) in SHAPES:
    ldy = rest[0] if rest else Cout  # SHAPE=mode,H,Cin,Cout,ldy: Cout gradient rows of an ldy-wide dY (the output conv: 65 of 128)
    Ho = H // 2 if mode == S2 else H
    taps = 1 if mode == X1 else 9
    x = torch.randn(B * H * H, Cin, device=dev).bfloat16()
    y = torch.randn(B * Ho * Ho, ldy, device=dev).bfloat16()
    dws = [torch.zeros(Cout * taps * Cin, device=dev) for _ in libs]
    a = ConvArgs(x.data_ptr(), None, None, None, None, y.data_ptr(), None, B, H, H, Cin, Ho, Ho, Cout, ldy, Cout, mode, 0, 0)
    gf = 2.0 * B * Ho * Ho * Cout * taps * Cin / 1e9

    def run(lib, dw):
        if os.environ.get("WS", "1") == "0":  # no workspace: split-K partial sums by fp32 atomics
            rc = lib.c2w_conv_wgrad(ctypes.byref(a), ctypes.c_void_p(dw.data_ptr()), None, None, 0, 1, st)
        else:
            rc = lib.c2w_conv_wgrad(ctypes.byref(a), ctypes.c_void_p(dw.data_ptr()), None, ctypes.c_void_p(ws.data_ptr()), ws.numel() * 4, 1, st)
        assert rc == 0, rc

    times = [[] for _ in libs]
    for (_, lib), dw in zip(libs, dws):
        for _ in range(3):
            run(lib, dw)
    torch.cuda.synchronize()
    for r in range(ROUNDS):
        for i, ((_, lib), dw) in enumerate(zip(libs, dws)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run(lib, dw)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 10 * 1e3)
    ref = dws[0]
    line = f"mode={mode} H={H:3d} {Cin:4d}->{Cout:4d}:"
    for i, (name, _) in enumerate(libs):
        med, mn = statistics.median(times[i]), min(times[i])
        err = (dws[i] - ref).abs().max().item() / ref.abs().max().item()
        line += f"  [{name}] med {med:7.1f} us min {mn:7.1f} us {gf / med * 1e3:7.1f} TF/s (diff vs first {err:.1e})"
    print(line, flush=True)
