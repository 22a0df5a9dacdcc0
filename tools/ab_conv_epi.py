"""Times the fused-epilogue flavours of the dominant forward / input-gradient launch (3x3, 128 -> 128 @128^2, B = 128, bf16) for one or
more builds of the library.  Each build runs in its own process (C2W_LIB), twice, in the order A B A B; per flavour the median of
ROUNDS x 10 launches.

    python tools/ab_conv_epi.py climate2weather_amd/build/alt/libc2w_base.so climate2weather_amd/build/alt/libc2w_epi2.so
"""
import math, os, subprocess, sys, statistics
sys.path.insert(0, os.getcwd())

if os.environ.get("_AB_CHILD"):
    import torch
    from climate2weather_amd import ops
    dev = torch.device("cuda:0")
    B, H, C = int(os.environ.get("B", "128")), int(os.environ.get("H", "128")), 128
    ROUNDS = int(os.environ.get("ROUNDS", "5"))
    T = torch.bfloat16
    g = dict(B=B, Hin=H, Win=H, Cin=C, Hout=H, Wout=H, Cout=C, ldy=C, wrows=C, mode=ops.CONV_S1)
    x = torch.randn(B * H * H, C, device=dev).to(T)
    w = (torch.randn(C, 9, C, device=dev) / math.sqrt(9 * C)).to(T)
    bias = torch.randn(C, device=dev)
    y = torch.empty(B * H * H, C, device=dev, dtype=T)
    y2 = torch.empty_like(y)
    res = torch.randn(B * H * H, C, device=dev).to(T)
    m = torch.randn(B, C, device=dev)
    dm = torch.zeros(B, C, device=dev)
    cases = (("bias", dict(bias=bias)), ("bias+silu", dict(bias=bias, act=ops.ACT_SILU)), ("bias+silu pair", dict(bias=bias, act=ops.ACT_SILU_PAIR, y2=y2)),
             ("bias+res", dict(bias=bias, res=res)), ("mul+res", dict(bias=bias, mul=res, res=res, mulmode=ops.MUL_DSILU)),
             ("res+LN fwd", dict(bias=bias, res=res, lnf=dict(y=y2, m=m, ldm=C, eps=1e-5, unbiased=True))),
             ("res+LN bwd", dict(res=res, ln=dict(x=res, m=m, dm=dm, ldm=C, eps=1e-5, unbiased=True))))
    out = []
    for name, kw in cases:
        kw = dict(kw)
        b_ = kw.pop("bias", None)
        try:
            for _ in range(3):
                ops.conv(x, w, b_, y, g, ops.DTYPE_BF16, **kw)
        except Exception as e:  # a flavour this build does not know
            out.append(f"{name}: {type(e).__name__}")
            continue
        torch.cuda.synchronize()
        ts = []
        for _ in range(ROUNDS):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv(x, w, b_, y, g, ops.DTYPE_BF16, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)
        out.append(f"{name} {statistics.median(ts):6.1f}")
    print("  ".join(out), flush=True)
    sys.exit(0)

libs = sys.argv[1:]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ, _AB_CHILD="1", C2W_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
        print(f"[{os.path.basename(lib)}] us: {r.stdout.strip() or r.stderr.strip()[-400:]}", flush=True)
