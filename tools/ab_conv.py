"""A/B of two (or more) builds of the library on forward / input-gradient implicit-GEMM launches (c2w_conv_forward) of the default
network (B = 128, bf16): interleaved rounds in ONE process on one device, median and min per build, difference of the results.

    python tools/ab_conv.py climate2weather_amd/build/libc2w_old.so climate2weather_amd/libc2w_hip.so
    SHAPES=0,1 ACT=1 python tools/ab_conv.py ...     (ACT: 0 none, 1 SiLU; BIAS=0 drops the bias: an input-gradient launch)
"""
import ctypes, math, os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd import _lib
from climate2weather_amd._lib import ConvArgs, c_int, c_longlong

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "128"))
ROUNDS = int(os.environ.get("ROUNDS", "7"))
ACT = int(os.environ.get("ACT", "0"))
BIAS = int(os.environ.get("BIAS", "1"))
F16 = int(os.environ.get("F16", "0"))  # 1: fp16 operands (same MFMA rate; some experiment builds only fit the fp16 instantiation)
TD = torch.float16 if F16 else torch.bfloat16
S1, S2, UP, TS2, X1 = _lib.CONV_S1, _lib.CONV_S2, _lib.CONV_UP, _lib.CONV_TS2, _lib.CONV_1X1


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
          (S1, 16, 512, 384), (S2, 128, 128, 128), (S2, 64, 128, 256), (S2, 32, 256, 384), (S2, 16, 384, 512), (TS2, 64, 128, 128), (TS2, 32, 256, 128),
          (TS2, 16, 384, 256), (TS2, 8, 512, 384), (X1, 8, 512, 1536), (X1, 8, 512, 512)]
if os.environ.get("SHAPE"):  # one custom geometry: SHAPE=mode,H,Cin,Cout[,ldy] (Cout output channels in rows of ldy: the output conv is 65 of 128)
    SHAPES = [tuple(int(v) for v in os.environ["SHAPE"].split(","))]
elif os.environ.get("SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["SHAPES"].split(",")]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (mode, H, Cin, Cout, *rest) in SHAPES:
    ldy = rest[0] if rest else Cout
    wrows = rest[1] if len(rest) > 1 else Cout  # SHAPE=1,128,128,128,128,65: the output conv (65 weight rows, 128-wide output rows)
    Ho = H // 2 if mode == S2 else (H * 2 if mode in (UP, TS2) else H)
    taps = 1 if mode == X1 else 9
    x = torch.randn(B * H * H, Cin, device=dev).to(TD)
    w = (torch.randn(rest[1] if len(rest) > 1 else Cout, taps, Cin, device=dev) / math.sqrt(taps * Cin)).to(TD)
    if os.environ.get("ZERO"):  # all-zero operands: same instruction stream, far less switching energy -> shows what the clock governor takes
        x.zero_()
        w.zero_()
    bias = torch.randn(rest[1] if len(rest) > 1 else Cout, device=dev)
    ys = [torch.zeros(B * Ho * Ho, ldy, device=dev, dtype=TD) for _ in libs]
    mac_pix = B * Ho * Ho if mode != TS2 else B * H * H
    gf = 2.0 * mac_pix * Cout * taps * Cin / 1e9

    def run(lib, y):
        a = ConvArgs(x.data_ptr(), w.data_ptr(), bias.data_ptr() if BIAS else None, None, None, y.data_ptr(), None, B, H, H, Cin, Ho, Ho, Cout, ldy, wrows,
                     mode, ACT, 0)
        rc = lib.c2w_conv_forward(ctypes.byref(a), _lib.DTYPE_F16 if F16 else _lib.DTYPE_BF16, 0, st)
        assert rc == 0, rc

    times = [[] for _ in libs]
    for (_, lib), y in zip(libs, ys):
        for _ in range(3):
            run(lib, y)
    torch.cuda.synchronize()
    for r in range(ROUNDS):
        for i, ((_, lib), y) in enumerate(zip(libs, ys)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run(lib, y)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 10 * 1e3)
    ref = ys[0].float()
    line = f"mode={mode} H={H:3d} {Cin:4d}->{Cout:4d}:"
    for i, (name, _) in enumerate(libs):
        med, mn = statistics.median(times[i]), min(times[i])
        err = (ys[i].float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
        line += f"\n   [{name}] med {med:7.1f} us min {mn:7.1f} us {gf / med * 1e3:7.1f} TF/s (diff vs first {err:.1e})"
    print(line, flush=True)
