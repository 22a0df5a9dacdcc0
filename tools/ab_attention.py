"""A/B of library builds on the bottleneck attention (T = 64 tokens, one head of width C, B images): forward and backward, interleaved
rounds in one process.   python tools/ab_attention.py old.so new.so"""
import ctypes, os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd import _lib
from climate2weather_amd._lib import c_int, c_longlong

dev = torch.device("cuda:0")
B, T, C = int(os.environ.get("B", "128")), 64, int(os.environ.get("C", "512"))
ROUNDS = int(os.environ.get("ROUNDS", "7"))


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, argtypes in _lib._PROTOS.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes = argtypes
            fn.restype = c_longlong if name.endswith("_bytes") else c_int
    return lib


libs = [(os.path.basename(p), load(p)) for p in sys.argv[1:]]
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * C, device=dev).bfloat16()
do = torch.randn(B * T, C, device=dev).bfloat16()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
outs = []
for name, lib in libs:
    o = torch.empty(B * T, C, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B * T, device=dev)
    dq = torch.empty_like(qkv)
    delta = torch.empty(B * T, device=dev)
    outs.append((o, lse, dq, delta))


def fwd(lib, o, lse):
    assert lib.c2w_attention_forward(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, T, C, _lib.DTYPE_BF16, st) == 0


def bwd(lib, o, lse, dq, delta):
    assert lib.c2w_attention_backward(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), B, T, C, _lib.DTYPE_BF16, st) == 0


tf, tb = [[] for _ in libs], [[] for _ in libs]
for (name, lib), (o, lse, dq, delta) in zip(libs, outs):
    for _ in range(3):
        fwd(lib, o, lse); bwd(lib, o, lse, dq, delta)
torch.cuda.synchronize()
for r in range(ROUNDS):
    for i, ((name, lib), (o, lse, dq, delta)) in enumerate(zip(libs, outs)):
        for times, fn in ((tf[i], lambda: fwd(lib, o, lse)), (tb[i], lambda: bwd(lib, o, lse, dq, delta))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / 20 * 1e3)
for i, (name, _) in enumerate(libs):
    print(f"[{name}] forward {statistics.median(tf[i]):6.1f} us  backward {statistics.median(tb[i]):6.1f} us   "
          f"max |o - o0| {(outs[i][0].float() - outs[0][0].float()).abs().max().item():.2e}  max |dqkv - dqkv0| {(outs[i][2].float() - outs[0][2].float()).abs().max().item():.2e}")
