"""GPU parity of every HIP kernel (called through the C ABI via climate2weather_amd.ops) against the plain PyTorch
restatement of the same op (tests/emu_ops.py) on the same seeded inputs.

Tolerances (written here, per SURVEY 8d / BASELINE.json): fp32 mode <= 1e-4 of the output scale (the parity gate);
bf16 mode <= 2e-2 of the output scale (bf16 has 8 significand bits; K up to 4608 products are accumulated in fp32);
fp16 mode (the reference's autocast type; 11 significand bits) <= 4e-3.
"""
import math

import numpy as np
import pytest
import torch

import emu_ops as E
from climate2weather_amd import _lib, ops

pytestmark = pytest.mark.gpu

F32, BF16, F16 = ops.DTYPE_F32, ops.DTYPE_BF16, ops.DTYPE_F16
TD = ops.TORCH_DTYPE
TOL = {F32: 1e-4, BF16: 2e-2, F16: 4e-3}


def dev():
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _knobs_follow_the_environment():
    """monkeypatch restores the environment after a test; the library's knob table (read once) must follow."""
    yield
    ops.knobs_reload()


def rnd(shape, dt, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(TD[dt]).to(dev())


def close(a, b, dt, what="", scale=None, tol=None):
    a, b = a.float(), b.float()
    s = scale if scale is not None else max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item()
    assert math.isfinite(err) and err <= (tol or TOL[dt]) * s, f"{what}: max err {err:.3e} vs scale {s:.3e}"


def geom(B, Hin, Win, Cin, Hout, Wout, Cout, ldy, wrows, mode):
    return dict(B=B, Hin=Hin, Win=Win, Cin=Cin, Hout=Hout, Wout=Wout, Cout=Cout, ldy=ldy, wrows=wrows, mode=mode)


CONV_CASES = [
    # (mode, B, Hin, Win, Cin, Cout, wrows, ldy)
    (ops.CONV_S1, 2, 16, 16, 64, 128, 128, 128),
    (ops.CONV_S1, 3, 8, 8, 128, 192, 192, 192),     # ragged pixel tile (192 px), two channel tiles, second partial
    (ops.CONV_S1, 1, 32, 32, 128, 64, 52, 64),      # output conv: 52 real rows, padded output rows stay zero
    (ops.CONV_S1, 2, 4, 4, 64, 64, 64, 64),         # tiny spatial (plumbing config depth)
    (ops.CONV_S1, 2, 32, 48, 128, 128, 128, 128),   # halo-patch kernel: 2x3 tiles per image, 2 K-chunks (bf16) / 4 (fp32)
    (ops.CONV_S1, 1, 16, 32, 192, 320, 320, 320),   # halo-patch kernel: odd chunk count, 3 channel tiles (last partial)
    (ops.CONV_S1, 2, 8, 16, 128, 128, 128, 128),    # halo-patch kernel: one 8x16 tile per image
    (ops.CONV_S1, 1, 24, 32, 64, 128, 128, 128),    # halo-patch kernel: height a multiple of 8 only
    (ops.CONV_S2, 2, 16, 16, 64, 128, 128, 128),
    (ops.CONV_S2, 1, 32, 32, 128, 256, 256, 256),
    (ops.CONV_UP, 2, 8, 8, 128, 64, 64, 64),
    (ops.CONV_UP, 1, 16, 16, 64, 128, 128, 128),
    (ops.CONV_UP, 2, 8, 16, 128, 128, 128, 128),    # upsampling folded into the halo patch: 16x32 output = 2x2 tiles of 8x16, 2 K-chunks (bf16)
    (ops.CONV_UP, 3, 12, 8, 64, 192, 180, 192),     # 24x16 output: three tiles per image, two channel tiles (second partial)
    (ops.CONV_TS2, 2, 8, 8, 128, 64, 64, 64),       # dgrad of a stride-2 conv: 8x8 dy -> 16x16 dx
    (ops.CONV_TS2, 4, 8, 8, 128, 64, 64, 64),       # same, a quarter of the pixels fills a tile: parity-class tiles (1/2/2/4 taps)
    (ops.CONV_TS2, 8, 16, 8, 64, 192, 192, 192),    # parity classes: 4 tiles per class, 2 images per tile, 2 channel tiles
    (ops.CONV_TS2, 2, 16, 16, 128, 128, 128, 128),  # parity-class halo-patch kernel: 2 tiles per image, 2 K-chunks (bf16)
    (ops.CONV_TS2, 3, 8, 32, 64, 192, 180, 192),    # same: two tiles per row, two channel tiles (second partial), masked weight rows
    (ops.CONV_1X1, 5, 1, 1, 64, 320, 320, 320),     # Linear on 5 rows
    (ops.CONV_1X1, 2, 8, 8, 128, 384, 384, 384),    # qkv-style 1x1
]


def _out_hw(mode, Hin, Win):
    if mode == ops.CONV_S2:
        return Hin // 2, Win // 2
    if mode in (ops.CONV_UP, ops.CONV_TS2):
        return Hin * 2, Win * 2
    return Hin, Win


@pytest.mark.parametrize("dt", [F32, BF16, F16])
@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("naive", [0, 1, 2])
def test_conv_forward(case, dt, naive):
    mode, B, Hin, Win, Cin, Cout, wrows, ldy = case
    taps = 1 if mode == ops.CONV_1X1 else 9
    Hout, Wout = _out_hw(mode, Hin, Win)
    g = geom(B, Hin, Win, Cin, Hout, Wout, Cout, ldy, wrows, mode)
    x = rnd((B * Hin * Win, Cin), dt, 1)
    w = rnd((wrows, taps, Cin), dt, 2, scale=1.0 / math.sqrt(taps * Cin))
    bias = rnd((wrows,), F32, 3)
    res = rnd((B * Hout * Wout, ldy), dt, 4)
    mul = rnd((B * Hout * Wout, ldy), dt, 5)
    for variant in range(5):
        kw = [dict(), dict(act=ops.ACT_SILU), dict(res=res), dict(mul=mul, mulmode=ops.MUL_DSILU, res=res), dict()][variant]
        b = None if variant == 3 else bias
        y = torch.full((B * Hout * Wout, ldy), 7.0, dtype=TD[dt], device=dev())
        y_ref = y.clone()
        y2 = torch.full_like(y, 3.0) if variant == 4 else None  # dual output: pre-activation + silu
        y2_ref = y2.clone() if variant == 4 else None
        ops.conv(x, w, b, y, g, dt, naive=naive, y2=y2, **kw)
        E.conv(x, w, b, y_ref, g, dt, y2=y2_ref, **kw)
        torch.cuda.synchronize()
        close(y, y_ref, dt, f"conv mode={mode} variant={variant} naive={naive}")
        if variant == 4:
            close(y2, y2_ref, dt, f"conv second output mode={mode} naive={naive}")


@pytest.mark.parametrize("dt", [BF16, F16])
def test_conv_16x16_tile_kernel_all_epilogues(dt):
    """Launches with >= 1024 workgroups take conv_patch_t3_kernel<16> (conv_patch3.hip); every epilogue variant of it against the
    PyTorch restatement at a size that dispatches there (B = 16, 128x128, 128 -> 128: 1024 tiles), 2 K-chunks."""
    B, H, W, C = 16, 128, 128, 128
    g = geom(B, H, W, C, H, W, C, C, C, ops.CONV_S1)
    npix = B * H * W
    x = rnd((npix, C), dt, 1)
    w = rnd((C, 9, C), dt, 2, scale=1.0 / math.sqrt(9 * C))
    bias = rnd((C,), F32, 3)
    res = rnd((npix, C), dt, 4)
    mul = rnd((npix, C), dt, 5)
    m = rnd((B, C + 64), F32, 6)
    y, y2 = (torch.full((npix, C), 7.0, dtype=TD[dt], device=dev()) for _ in range(2))
    y_ref, y2_ref = y.clone(), y2.clone()
    variants = [dict(bias=bias), dict(bias=bias, act=ops.ACT_SILU), dict(bias=bias, res=res), dict(mul=mul, mulmode=ops.MUL_DSILU, res=res),
                dict(mul=mul, mulmode=ops.MUL_PLAIN), dict(bias=bias, y2=True), dict(bias=bias, act=ops.ACT_SILU_PAIR, y2=True)]
    for kw in variants:
        kw = dict(kw)
        b = kw.pop("bias", None)
        two = kw.pop("y2", False)
        ops.conv(x, w, b, y, g, dt, y2=y2 if two else None, **kw)
        E.conv(x, w, b, y_ref, g, dt, y2=y2_ref if two else None, **kw)
        torch.cuda.synchronize()
        close(y, y_ref, dt, f"16x16 tile kernel {sorted(kw)}")
        if two:
            close(y2, y2_ref, dt, f"16x16 tile kernel second output {sorted(kw)}")
    dm, dm_ref = torch.zeros_like(m), torch.zeros_like(m)
    ln = dict(x=mul, m=m.view(-1)[32:], ldm=C + 64, eps=1e-5, unbiased=True)
    ops.conv(x, w, None, y, g, dt, res=res, ln=dict(ln, dm=dm.view(-1)[32:]))
    E.conv(x, w, None, y_ref, g, dt, res=res, ln=dict(ln, dm=dm_ref.view(-1)[32:]))
    close(y, y_ref, dt, "16x16 tile kernel, fused LN backward")
    close(dm, dm_ref, dt, "16x16 tile kernel, fused LN backward dm", tol=1e-2)
    lnf = dict(m=m.view(-1)[32:], ldm=C + 64, eps=1e-5, unbiased=True)
    ops.conv(x, w, bias, y, g, dt, res=res, lnf=dict(lnf, y=y2))
    E.conv(x, w, bias, y_ref, g, dt, res=res, lnf=dict(lnf, y=y2_ref))
    close(y, y_ref, dt, "16x16 tile kernel next to fused LN forward")
    close(y2, y2_ref, dt, "16x16 tile kernel, fused LN forward output")


@pytest.mark.parametrize("dt", [F32, BF16, F16])
@pytest.mark.parametrize("shape", [(2, 16, 32, 64, 128), (3, 24, 16, 128, 192), (1, 8, 16, 64, 64), (16, 128, 128, 128, 128)])
def test_conv_with_pooled_output(shape, dt):
    """C2W_CONV_POOL2: the input gradient of the up-conv leaves the kernel as 2x2 sums (the adjoint of Upsample(nearest, x2),
    model/nn.py:184) -- halo-patch kernels only.  8x16-tile kernel (small shapes, all dtypes) and the 16x16-tile kernel (B = 16 at
    128x128: 1024 tiles) against conv + sumpool2 of the PyTorch restatement, and bit-equal to the kernels' own two-pass result."""
    B, H, W, Cin, Cout = shape
    if dt == F32 and B == 16:
        pytest.skip("the 16x16-tile kernel is 16-bit only; fp32 covered by the small shapes")
    g = geom(B, H, W, Cin, H, W, Cout, Cout, Cout, ops.CONV_S1)
    assert ops.conv_pool2_supported(g, dt)
    x = rnd((B * H * W, Cin), dt, 1)
    w = rnd((Cout, 9, Cin), dt, 2, scale=1.0 / math.sqrt(9 * Cin))
    yp = torch.full((B * H * W // 4, Cout), 7.0, dtype=TD[dt], device=dev())
    yp_ref = yp.clone()
    ops.conv(x, w, None, yp, g, dt, pool2=True)
    E.conv(x, w, None, yp_ref, g, dt, pool2=True)
    close(yp, yp_ref, dt, "pooled conv output")
    full = torch.empty((B * H * W, Cout), dtype=TD[dt], device=dev())
    two = torch.empty_like(yp)
    ops.conv(x, w, None, full, g, dt)
    ops.sumpool2(full, two, B, H // 2, W // 2, Cout, dt)
    assert torch.equal(yp, two)  # same values in the same order as the separate pooling pass
    assert not ops.conv_pool2_supported(geom(B, H, W, Cin, H // 2, W // 2, Cout, Cout, Cout, ops.CONV_S2), dt)
    with pytest.raises(_lib.C2wError):
        ops.conv(x, w, None, yp, geom(B, H, W, Cin, H // 2, W // 2, Cout, Cout, Cout, ops.CONV_S2), dt, pool2=True)


@pytest.mark.parametrize("dt", [BF16, F16])
def test_up_conv_on_the_16x16_tile_kernel_and_its_weight_gradient(dt):
    """model/nn.py:184-189 (LayerNorm -> Upsample(nearest, x2) -> Conv2d) without the upsampled map: C2W_CONV_UP on the halo-patch
    kernels fetches patch pixel (ih, iw) from source pixel (ih >> 1, iw >> 1).  At a size that dispatches to conv_patch_t3_kernel<16>
    (B = 16, 64x64 -> 128x128, 128 -> 128: 1024 tiles) with the epilogues the engine uses there (skip add, next block's LayerNorm
    emitted), and the weight gradient of the same geometry on wgrad_patch_kernel -- against the PyTorch restatement."""
    B, Hl, C = 16, 64, 128
    H = 2 * Hl
    g = geom(B, Hl, Hl, C, H, H, C, C, C, ops.CONV_UP)
    assert ops.conv_patch_supported(g, dt) and ops.conv_lnfwd_supported(g, dt)
    x = rnd((B * Hl * Hl, C), dt, 1)
    w = rnd((C, 9, C), dt, 2, scale=1.0 / math.sqrt(9 * C))
    bias = rnd((C,), F32, 3)
    res = rnd((B * H * H, C), dt, 4)
    m = rnd((B, C + 64), F32, 6)
    y, y2 = (torch.full((B * H * H, C), 7.0, dtype=TD[dt], device=dev()) for _ in range(2))
    y_ref, y2_ref = y.clone(), y2.clone()
    ops.conv(x, w, bias, y, g, dt, res=res)
    E.conv(x, w, bias, y_ref, g, dt, res=res)
    close(y, y_ref, dt, "up-conv + skip on the 16x16 tile kernel")
    lnf = dict(m=m.view(-1)[32:], ldm=C + 64, eps=1e-5, unbiased=True)
    ops.conv(x, w, bias, y, g, dt, res=res, lnf=dict(lnf, y=y2))
    E.conv(x, w, bias, y_ref, g, dt, res=res, lnf=dict(lnf, y=y2_ref))
    close(y, y_ref, dt, "up-conv next to fused LN forward")
    close(y2, y2_ref, dt, "up-conv, fused LN forward output")
    dy = rnd((B * H * H, C), dt, 7)
    dw = torch.zeros(C * 9 * C, dtype=torch.float32, device=dev())
    db = torch.zeros(C, dtype=torch.float32, device=dev())
    dw_ref, db_ref = dw.clone(), db.clone()
    ops.conv_wgrad(x, dy, dw, g, dt, dbias=db, workspace=ops.new_workspace(dev()))
    E.conv_wgrad(x, dy, dw_ref, g, dt, dbias=db_ref)
    close(dw, dw_ref, dt, "up-conv weight gradient (halo patch from the low-resolution map)", tol=1e-2)
    close(db, db_ref, dt, "up-conv bias gradient", tol=1e-2)


@pytest.mark.parametrize("dt", [F32, BF16, F16])
@pytest.mark.parametrize("naive", [0, 1, 2])
@pytest.mark.parametrize("case", [(ops.CONV_S1, 2, 16, 16, 64, 128, 128, 128), (ops.CONV_S1, 3, 8, 8, 128, 192, 192, 192),
                                  (ops.CONV_S1, 1, 24, 32, 64, 128, 128, 128)])
def test_conv_silu_pair_outputs(case, dt, naive):
    """training epilogue of a res-block's first conv: y = silu(a), y2 = silu'(a) (C2W_ACT_SILU_PAIR), and the plain
    multiplier the backward pass then uses"""
    mode, B, Hin, Win, Cin, Cout, wrows, ldy = case
    g = geom(B, Hin, Win, Cin, Hin, Win, Cout, ldy, wrows, mode)
    x = rnd((B * Hin * Win, Cin), dt, 1)
    w = rnd((wrows, 9, Cin), dt, 2, scale=1.0 / math.sqrt(9 * Cin))
    bias = rnd((wrows,), F32, 3)
    y, y2 = (torch.full((B * Hin * Win, ldy), 7.0, dtype=TD[dt], device=dev()) for _ in range(2))
    y_ref, y2_ref = y.clone(), y2.clone()
    ops.conv(x, w, bias, y, g, dt, act=ops.ACT_SILU_PAIR, y2=y2, naive=naive)
    E.conv(x, w, bias, y_ref, g, dt, act=ops.ACT_SILU_PAIR, y2=y2_ref)
    torch.cuda.synchronize()
    close(y, y_ref, dt, "silu output")
    close(y2, y2_ref, dt, "silu' output")
    dx, dx_ref = torch.empty_like(y), torch.empty_like(y)
    wT = rnd((Cout, 9, Cout), dt, 4, scale=1.0 / math.sqrt(9 * Cout))
    gd = geom(B, Hin, Win, ldy, Hin, Win, Cout, ldy, Cout, mode)
    ops.conv(y, wT, None, dx, gd, dt, mul=y2, mulmode=ops.MUL_PLAIN, naive=naive)
    E.conv(y, wT, None, dx_ref, gd, dt, mul=y2, mulmode=ops.MUL_PLAIN)
    close(dx, dx_ref, dt, "plain multiplier epilogue")


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("unbiased", [True, False])
@pytest.mark.parametrize("B,H,W,Cin,per_sample,use_res", [(2, 32, 48, 128, True, True), (3, 8, 16, 192, True, False), (1, 16, 32, 64, False, True)])
def test_conv_with_fused_ln_forward_output(B, H, W, Cin, per_sample, use_res, unbiased, dt):
    """Second output of the conv epilogue: the consumer's LayerNorm input LN(y + m) (C2wConvArgs.lnf_*), against conv
    followed by ln_forward on the stored result."""
    C = 128
    g = geom(B, H, W, Cin, H, W, C, C, C, ops.CONV_S1)
    assert ops.conv_lnfwd_supported(g, dt) and not ops.conv_lnfwd_supported(g, F32)
    npix = B * H * W
    x = rnd((npix, Cin), dt, 1)
    w = rnd((C, 9, Cin), dt, 2, scale=1.0 / math.sqrt(9 * Cin))
    bias = rnd((C,), F32, 3)
    res = rnd((npix, C), dt, 4) if use_res else None
    m = rnd((B if per_sample else 1, C + 64), F32, 5)
    ldm = C + 64 if per_sample else 0
    y, hn = (torch.full((npix, C), 7.0, dtype=TD[dt], device=dev()) for _ in range(2))
    y_ref, hn_ref = y.clone(), hn.clone()
    for mm in (m.view(-1)[32:], None):
        lnf = dict(m=mm, ldm=ldm if mm is not None else 0, eps=1e-5, unbiased=unbiased)
        ops.conv(x, w, bias, y, g, dt, res=res, lnf=dict(lnf, y=hn))
        E.conv(x, w, bias, y_ref, g, dt, res=res, lnf=dict(lnf, y=hn_ref))
        torch.cuda.synchronize()
        close(y, y_ref, dt, "conv output next to the fused LN")
        close(hn, hn_ref, dt, "fused LN forward output")
        # ... and, asked for, every pixel row's 1/sigma (C2wConvArgs.lnf_rstd), with the other outputs unchanged bit for bit
        rstd = torch.full((npix + 8,), -1.0, dtype=torch.float32, device=dev())
        rstd_ref = rstd.clone()
        y2_, hn2_ = torch.empty_like(y), torch.empty_like(hn)
        ops.conv(x, w, bias, y2_, g, dt, res=res, lnf=dict(lnf, y=hn2_, rstd=rstd))
        E.conv(x, w, bias, y_ref, g, dt, res=res, lnf=dict(lnf, y=hn_ref, rstd=rstd_ref))
        torch.cuda.synchronize()
        assert torch.equal(y2_, y) and torch.equal(hn2_, hn)
        close(rstd[:npix], rstd_ref[:npix], dt, "per-pixel 1/sigma of the fused LN", tol=2e-3)
        assert (rstd[npix:] == -1.0).all()


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("unbiased", [True, False])
@pytest.mark.parametrize("B,H,W,Cin,per_sample", [(2, 32, 48, 128, True), (3, 16, 16, 192, True), (1, 16, 32, 64, False)])
def test_conv_with_fused_ln_backward(B, H, W, Cin, per_sample, unbiased, dt):
    """The input-gradient conv of a res-block's first conv with LayerNorm's backward in its epilogue
    (C2wConvArgs.ln_*): y = res + dLN(conv(x); ln_x + m), dm += column sums -- against conv followed by ln_backward."""
    C = 128
    g = geom(B, H, W, Cin, H, W, C, C, C, ops.CONV_S1)
    assert ops.conv_lnbwd_supported(g, dt)
    assert not ops.conv_lnbwd_supported(g, F32)
    assert not ops.conv_lnbwd_supported(geom(B, H, W, Cin, H, W, 256, 256, 256, ops.CONV_S1), dt)
    npix = B * H * W
    x = rnd((npix, Cin), dt, 1)
    w = rnd((C, 9, Cin), dt, 2, scale=1.0 / math.sqrt(9 * Cin))
    lnx = rnd((npix, C), dt, 3)
    res = rnd((npix, C), dt, 4)
    ldm_total = C + 64
    m = rnd((B if per_sample else 1, ldm_total), F32, 5)
    ldm = ldm_total if per_sample else 0
    dm, dm_ref = torch.zeros_like(m), torch.zeros_like(m)
    y = torch.full((npix, C), 7.0, dtype=TD[dt], device=dev())
    y_ref = y.clone()
    ln = dict(x=lnx, m=m.view(-1)[32:], dm=dm.view(-1)[32:], ldm=ldm, eps=1e-5, unbiased=unbiased)
    ops.conv(x, w, None, y, g, dt, res=res, ln=ln)
    E.conv(x, w, None, y_ref, g, dt, res=res, ln=dict(ln, dm=dm_ref.view(-1)[32:]))
    torch.cuda.synchronize()
    close(y, y_ref, dt, "fused ln bwd dx")
    close(dm, dm_ref, dt, "fused ln bwd dm", tol=1e-2)
    # the same with the statistics the forward kept (C2wConvArgs.ln_rstd): ln_x = the normalised rows, ln_rstd their 1/sigma
    xm = lnx.float() + E._mrows(m.view(-1)[32:], npix, H * W, C, ldm)
    rs = (xm.var(dim=1, unbiased=unbiased) + 1e-5).rsqrt()
    xhat = ((xm - xm.mean(dim=1, keepdim=True)) * rs.unsqueeze(1)).to(TD[dt])
    dm2, dm2_ref = torch.zeros_like(m), torch.zeros_like(m)
    y_s, y_s_ref = torch.full_like(y, 7.0), torch.full_like(y, 7.0)
    lns = dict(x=xhat, rstd=rs.contiguous(), m=None, dm=dm2.view(-1)[32:], ldm=ldm, eps=1e-5, unbiased=unbiased)
    ops.conv(x, w, None, y_s, g, dt, res=res, ln=lns)
    E.conv(x, w, None, y_s_ref, g, dt, res=res, ln=dict(lns, dm=dm2_ref.view(-1)[32:]))
    torch.cuda.synchronize()
    close(y_s, y_s_ref, dt, "fused ln bwd dx from kept statistics")
    close(dm2, dm2_ref, dt, "fused ln bwd dm from kept statistics", tol=1e-2)
    close(y_s, y_ref, dt, "kept statistics vs recomputed (xhat rounded to the storage type)", tol=2 * TOL[dt])
    # without modulation / residual / dm
    ops.conv(x, w, None, y, g, dt, ln=dict(x=lnx, eps=1e-5, unbiased=unbiased))
    E.conv(x, w, None, y_ref, g, dt, ln=dict(x=lnx, eps=1e-5, unbiased=unbiased))
    close(y, y_ref, dt, "fused ln bwd dx (no m, no res)")


@pytest.mark.parametrize("dt", [F32, BF16, F16])
@pytest.mark.parametrize("force_gather", [False, True, "workspace"])
@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[0] != ops.CONV_TS2])
def test_conv_wgrad(case, dt, force_gather, monkeypatch):
    mode, B, Hin, Win, Cin, Cout, wrows, ldy = case
    ws = ops.new_workspace(dev()) if force_gather == "workspace" else None  # split-K partial sums through a scratch buffer instead of atomics
    if force_gather is True:
        if mode != ops.CONV_S1:
            pytest.skip("only 3x3 stride-1 has two kernels")
        monkeypatch.setenv("C2W_FORCE_GATHER", "1")  # the general kernel on shapes the halo-patch kernel would take
        ops.knobs_reload()  # the library reads its knobs once
    taps = 1 if mode == ops.CONV_1X1 else 9
    Hout, Wout = _out_hw(mode, Hin, Win)
    if force_gather is True:
        assert ops.conv_wgrad_dispatch(geom(B, Hin, Win, Cin, Hout, Wout, wrows, ldy, wrows, mode), dt) == _lib.KERNEL_GATHER
    Cw = wrows  # gradient rows = real output channels
    g = geom(B, Hin, Win, Cin, Hout, Wout, Cw, ldy, wrows, mode)
    x = rnd((B * Hin * Win, Cin), dt, 1)
    dy = rnd((B * Hout * Wout, ldy), dt, 2)
    dw = torch.zeros(Cw * taps * Cin + 64, dtype=torch.float32, device=dev())
    dw_ref = dw.clone()
    db = torch.zeros(Cw + 8, dtype=torch.float32, device=dev())
    db_ref = db.clone()
    ops.conv_wgrad(x, dy, dw, g, dt, dbias=db, workspace=ws)
    E.conv_wgrad(x, dy, dw_ref, g, dt, dbias=db_ref)
    torch.cuda.synchronize()
    need = ops.conv_wgrad_workspace_bytes(g, dt)
    assert 0 <= need <= ops.WORKSPACE_BYTES
    close(dw, dw_ref, dt, f"wgrad mode={mode}", tol=1e-4 if dt == F32 else 1e-2)
    close(db, db_ref, dt, f"wgrad bias mode={mode}", tol=1e-4 if dt == F32 else 1e-2)
    assert db[Cw:].abs().max().item() == 0.0
    assert dw[-64:].abs().max().item() == 0.0  # nothing written past the tensor
    # accumulation semantics: a second call adds
    ops.conv_wgrad(x, dy, dw, g, dt, workspace=ws)
    close(dw, 2 * dw_ref, dt, "wgrad accumulate", tol=1e-4 if dt == F32 else 1e-2)
    if ws is not None and need > 0:  # a buffer smaller than the launch needs is not used: same result through atomics
        small = torch.empty(max(need // 4 - 4, 4), dtype=torch.float32, device=dev())
        dw2 = torch.zeros_like(dw)
        ops.conv_wgrad(x, dy, dw2, g, dt, workspace=small)
        close(dw2, dw_ref, dt, "wgrad with an undersized workspace", tol=1e-4 if dt == F32 else 1e-2)


GROUP_CASES = [
    # (mode, B, H, W, Cin, Cout, layers)            split plan of the grouped launch (256 workgroups per round)
    (ops.CONV_S1, 4, 8, 8, 128, 256, 6),            # 8x8 images in pairs, 2 K tiles: no split, tiles added straight onto dw
    (ops.CONV_S1, 16, 8, 8, 256, 256, 12),          # pairs, 8 K tiles, 96 output tiles: two splits
    (ops.CONV_S1, 6, 16, 16, 128, 128, 3),          # 12 K tiles, 6 output tiles: split 12 ways
    (ops.CONV_S1, 2, 32, 32, 128, 192, 2),          # partial second channel tile
    (ops.CONV_S1, 2, 16, 32, 64, 128, 16),          # the most layers one launch takes
    (ops.CONV_UP, 2, 8, 16, 128, 128, 2),           # upsampling folded into the patch load
    (ops.CONV_1X1, 640, 1, 1, 128, 384, 6),         # attention qkv (model/nn.py:45) on the gather kernel: 10 K stages, 3 x 1 tiles, split
    (ops.CONV_1X1, 96, 1, 1, 256, 256, 3),          # 2 K stages: no split (tiles added onto dw by single-adder atomics)
]


@pytest.mark.parametrize("dt", [F32, BF16, F16])
@pytest.mark.parametrize("case", GROUP_CASES)
def test_conv_wgrad_grouped(case, dt):
    """c2w_conv_wgrad_grouped: the weight gradients of n layers of one geometry in one launch == n c2w_conv_wgrad calls (another split of
    the same pixel sum: equal to rounding), accumulates onto dw / dbias like them, writes nothing else, and is bit-reproducible."""
    mode, B, Hin, Win, Cin, Cout, n = case
    if dt == F32:
        Cin, Cout = Cin // 2 if Cin > 64 else Cin, Cout // 2  # fp32 tiles are half as wide: same number of tiles
    Hout, Wout = _out_hw(mode, Hin, Win)
    g = geom(B, Hin, Win, Cin, Hout, Wout, Cout, Cout, Cout, mode)
    assert ops.conv_wgrad_grouped_supported(g, n, dt)
    assert not ops.conv_wgrad_grouped_supported(g, 1, dt) and not ops.conv_wgrad_grouped_supported(g, 17, dt)
    xs = [rnd((B * Hin * Win, Cin), dt, 10 + i) for i in range(n)]
    dys = [rnd((B * Hout * Wout, Cout), dt, 40 + i) for i in range(n)]
    taps = 1 if mode == ops.CONV_1X1 else 9
    pre = [rnd((Cout * taps * Cin + 64,), F32, 70 + i, 0.1) for i in range(n)]  # dw holds a value already: the call accumulates
    for t in pre:
        t[-64:] = 0
    dws = [t.clone() for t in pre]
    dbs = [torch.zeros(Cout + 8, dtype=torch.float32, device=dev()) for _ in range(n)]
    need = ops.conv_wgrad_grouped_workspace_bytes(g, n, dt)
    ws = torch.empty(max(need // 4, 4), dtype=torch.float32, device=dev())
    ops.conv_wgrad_grouped([(xs[i], dys[i], dws[i], dbs[i] if i % 2 == 0 else None) for i in range(n)], g, dt, workspace=ws)
    one_ws = ops.new_workspace(dev())
    for i in range(n):
        dw_ref, db_ref = pre[i].clone(), torch.zeros_like(dbs[i])
        ops.conv_wgrad(xs[i], dys[i], dw_ref, g, dt, dbias=db_ref, workspace=one_ws)
        emu = pre[i].clone()
        E.conv_wgrad(xs[i], dys[i], emu, g, dt)
        close(dws[i], emu, dt, f"grouped wgrad layer {i} vs restatement", tol=1e-4 if dt == F32 else 1e-2)
        close(dws[i], dw_ref, dt, f"grouped wgrad layer {i} vs single launch", tol=1e-5 if dt == F32 else 2e-3)
        if i % 2 == 0:
            close(dbs[i], db_ref, dt, f"grouped wgrad bias {i}", tol=1e-4 if dt == F32 else 1e-2)
        else:
            assert dbs[i].abs().max().item() == 0.0
        assert dws[i][-64:].abs().max().item() == 0.0 and dbs[i][Cout:].abs().max().item() == 0.0
    again = [t.clone() for t in pre]
    ops.conv_wgrad_grouped([(xs[i], dys[i], again[i], None) for i in range(n)], g, dt, workspace=ws)
    torch.cuda.synchronize()
    for i in range(n):
        assert torch.equal(again[i], dws[i]), f"layer {i}: two grouped launches of the same inputs differ"
    if need > 0:  # a workspace smaller than the plan needs is refused (the caller sizes it with ..._workspace_bytes), nothing is written
        with pytest.raises(_lib.C2wError):
            ops.conv_wgrad_grouped([(xs[i], dys[i], again[i], None) for i in range(n)], g, dt, workspace=ws[: need // 4 - 4])


@pytest.mark.parametrize("dt", [F32, BF16, F16])
@pytest.mark.parametrize("shape", [(2, 64, 128, True), (3, 16, 256, True), (1, 1024, 384, False), (2, 64, 512, True), (4, 4, 64, True)])
def test_layernorm_fwd_bwd(shape, dt):
    B, HW, C, per_sample = shape
    if dt == F32 and C > 512:
        pytest.skip("fp32 rows hold at most 512 channels in registers")
    npix = B * HW
    x = rnd((npix, C), dt, 1)
    ldm_total = C + 64
    m = rnd((B if per_sample else 1, ldm_total), F32, 2)
    ldm = ldm_total if per_sample else 0
    for use_m in (True, False):
        mm = m.view(-1)[32:] if use_m else None
        y = torch.empty_like(x)
        y_ref = torch.empty_like(x)
        ops.ln_forward(x, mm, y, npix, HW, C, ldm, 1e-5, True, dt)
        E.ln_forward(x, mm, y_ref, npix, HW, C, ldm, 1e-5, True, dt)
        close(y, y_ref, dt, "ln fwd")
        dy = rnd((npix, C), dt, 3)
        dres = rnd((npix, C), dt, 4)
        dx, dx_ref = torch.empty_like(x), torch.empty_like(x)
        dm = torch.zeros_like(m) if use_m else None
        dm_ref = torch.zeros_like(m) if use_m else None
        ops.ln_backward(dy, x, mm, dres, dx, dm.view(-1)[32:] if use_m else None, npix, HW, C, ldm, 1e-5, True, dt)
        E.ln_backward(dy, x, mm, dres, dx_ref, dm_ref.view(-1)[32:] if use_m else None, npix, HW, C, ldm, 1e-5, True, dt)
        close(dx, dx_ref, dt, "ln bwd dx")
        if use_m:
            close(dm, dm_ref, dt, "ln bwd dm", tol=1e-4 if dt == F32 else 1e-2)
    # biased variant switch
    y = torch.empty_like(x)
    y_ref = torch.empty_like(x)
    ops.ln_forward(x, None, y, npix, HW, C, 0, 1e-5, False, dt)
    E.ln_forward(x, None, y_ref, npix, HW, C, 0, 1e-5, False, dt)
    close(y, y_ref, dt, "ln fwd biased")


@pytest.mark.parametrize("dt", [F32, BF16, F16])
def test_weight_transpose_batched(dt):
    """all input-gradient operands of a network in one launch == one launch per matrix"""
    specs = [(64, 9, 40, 64, 72, 1), (128, 1, 128, 128, 128, 0), (33, 9, 64, 64, 40, 1)]  # (R, NT, K, ldk, ldr, flip)
    flat = rnd((200000,), F32, 1)
    out = torch.zeros(400000, dtype=TD[dt], device=dev())
    out_ref = out.clone()
    desc, w_off, o_off = [], 0, 0
    for R, NT, K, ldk, ldr, flip in specs:
        desc += [w_off, o_off, R, NT, K, ldk, ldr, flip]
        ops.weight_transpose(flat[w_off:], out_ref[o_off:], R, NT, K, ldk, ldr, flip, dt)
        w_off += R * NT * ldk
        o_off += K * NT * ldr
    ops.weight_transpose_batched(flat, out, torch.tensor(desc, dtype=torch.int64, device=dev()), len(specs), dt)
    assert torch.equal(out, out_ref)


@pytest.mark.parametrize("dt", [F32, BF16, F16])
def test_upsample2(dt):
    B, H, W, C = 3, 5, 7, 64
    x = rnd((B * H * W, C), dt, 1)
    y = torch.empty((B * 4 * H * W, C), dtype=TD[dt], device=dev())
    y_ref = torch.empty_like(y)
    ops.upsample2(x, y, B, H, W, C, dt)
    E.upsample2(x, y_ref, B, H, W, C, dt)
    assert torch.equal(y, y_ref)
    assert ops.conv_patch_supported(geom(2, 8, 16, 64, 8, 16, 128, 128, 128, ops.CONV_S1), dt)
    assert not ops.conv_patch_supported(geom(2, 8, 8, 64, 8, 8, 128, 128, 128, ops.CONV_S1), dt)
    assert not ops.conv_patch_supported(geom(2, 16, 16, 64, 8, 8, 128, 128, 128, ops.CONV_S2), dt)


@pytest.mark.parametrize("dt", [F32, BF16, F16])
def test_pointwise_family(dt):
    d = dev()
    rows, C, lda = 1000, 128, 192
    a = rnd((rows, lda), dt, 1)
    out = torch.zeros(C + 8, device=d)
    out_ref = out.clone()
    ops.colsum(a, out, rows, C, lda, dt)
    E.colsum(a, out_ref, rows, C, lda, dt)
    close(out, out_ref, dt, "colsum", tol=1e-4 if dt == F32 else 1e-3)
    for rows2, C2 in ((5, 8448), (3000, 384), (70000, 64)):  # wide rows (modulation bias), odd vector counts, many blocks
        a2 = rnd((rows2, C2), dt, 7)
        o2 = torch.zeros(C2, device=d)
        o2_ref = o2.clone()
        ops.colsum(a2, o2, rows2, C2, C2, dt)
        E.colsum(a2, o2_ref, rows2, C2, C2, dt)
        close(o2, o2_ref, dt, f"colsum {rows2}x{C2}", tol=1e-4 if dt == F32 else 2e-3)
    n = 4096 * 8
    x = rnd((n,), dt, 2, scale=3.0)
    dy = rnd((n,), dt, 3)
    y, y_ref = torch.empty_like(x), torch.empty_like(x)
    ops.silu(x, y, n, dt)
    E.silu(x, y_ref, n, dt)
    close(y, y_ref, dt, "silu")
    ops.silu_backward(x, dy, y, n, dt)
    E.silu_backward(x, dy, y_ref, n, dt)
    close(y, y_ref, dt, "silu bwd")
    B, H, W, Cc = 2, 8, 8, 64
    g = rnd((B * 2 * H * 2 * W, Cc), dt, 4)
    p, p_ref = torch.empty((B * H * W, Cc), dtype=TD[dt], device=d), torch.empty((B * H * W, Cc), dtype=TD[dt], device=d)
    ops.sumpool2(g, p, B, H, W, Cc, dt)
    E.sumpool2(g, p_ref, B, H, W, Cc, dt)
    close(p, p_ref, dt, "sumpool2")


@pytest.mark.parametrize("dt", [F32, BF16, F16])
def test_layout_noise_loss(dt):
    d = dev()
    B, C, H, W, ldc = 3, 52, 16, 16, 64
    x = rnd((B, C, H, W), F32, 1)
    eps = rnd((B, C, H, W), F32, 2)
    t = torch.rand(B, generator=torch.Generator().manual_seed(3)).to(d)
    ms, ms_ref = torch.empty(B, 2, device=d), torch.empty(B, 2, device=d)
    ops.mu_sigma(t, ms, B, 1e-3)
    E.mu_sigma(t, ms_ref, B, 1e-3)
    close(ms, ms_ref, F32, "mu_sigma", tol=1e-6)
    for e in (None, eps):
        y = torch.full((B * H * W, ldc), 5.0, dtype=TD[dt], device=d)
        y_ref = y.clone()
        ops.nchw_to_nhwc(x, e, ms if e is not None else None, y, B, C, H * W, ldc, dt)
        E.nchw_to_nhwc(x, e, ms if e is not None else None, y_ref, B, C, H * W, ldc, dt)
        close(y, y_ref, dt, "nchw_to_nhwc")
        assert y[:, C:].abs().max().item() == 0.0
    back, back_ref = torch.empty_like(x), torch.empty_like(x)
    ops.nhwc_to_nchw(y, back, B, C, H * W, ldc, dt)
    E.nhwc_to_nchw(y_ref, back_ref, B, C, H * W, ldc, dt)
    close(back, back_ref, dt, "nhwc_to_nchw")
    dy, dy_ref = torch.empty_like(y), torch.empty_like(y)
    ls, ls_ref = torch.zeros(1, device=d), torch.zeros(1, device=d)
    ops.mse_loss_grad(y, eps, dy, ls, B, C, H * W, ldc, 0.37, dt)
    E.mse_loss_grad(y, eps, dy_ref, ls_ref, B, C, H * W, ldc, 0.37, dt)
    close(dy, dy_ref, dt, "mse dy")
    close(ls, ls_ref, F32, "mse loss", tol=1e-4)
    assert dy[:, C:].abs().max().item() == 0.0


def test_time_embedding_cast_transpose_adamw():
    d = dev()
    t = torch.tensor([0.0, 0.25, 0.5, 1.0], device=d)
    o, o_ref = torch.empty(4, 32, device=d), torch.empty(4, 32, device=d)
    ops.timestep_embedding(t, o, 4, 32)
    E.timestep_embedding(t, o_ref, 4, 32)
    close(o, o_ref, F32, "timestep_embedding", tol=1e-6)
    assert abs(o[2, 0].item() - 0.87758255) < 1e-6 and abs(o[2, 16].item() - 0.47942555) < 1e-6  # SURVEY a1 KAT
    src = rnd((10000,), F32, 1)
    for dt in (F32, BF16, F16):
        dst, dst_ref = torch.empty(10000, dtype=TD[dt], device=d), torch.empty(10000, dtype=TD[dt], device=d)
        ops.cast_f32(src, dst, 10000, dt)
        E.cast_f32(src, dst_ref, 10000, dt)
        assert torch.equal(dst, dst_ref)
        R, NT, K, ldk, ldr = 52, 9, 100, 128, 64
        w = rnd((R * NT * ldk,), F32, 2)
        for flip in (0, 1):
            out = torch.zeros(K * NT * ldr, dtype=TD[dt], device=d)
            out_ref = out.clone()
            ops.weight_transpose(w, out, R, NT, K, ldk, ldr, flip, dt)
            E.weight_transpose(w, out_ref, R, NT, K, ldk, ldr, flip, dt)
            assert torch.equal(out, out_ref)
    n = 5000
    bufs = [rnd((n,), F32, s) for s in (1, 2, 3)]
    p, g_, ema = bufs
    m = torch.zeros(n, device=d)
    v = torch.zeros(n, device=d)
    sh = torch.empty(n, dtype=torch.bfloat16, device=d)
    ref = [b.clone() for b in (p, g_, m, v, ema)]
    sh_ref = sh.clone()
    for step in (1, 2, 3):
        ops.adamw_ema(p, g_, m, v, ema, sh, n, 1e-3, 0.9, 0.999, 1e-8, 1e-3, step, 0.9999, 0.5)
        E.adamw_ema(ref[0], ref[1], ref[2], ref[3], ref[4], sh_ref, n, 1e-3, 0.9, 0.999, 1e-8, 1e-3, step, 0.9999, 0.5)
    close(p, ref[0], F32, "adamw p", tol=1e-6)
    close(ema, ref[4], F32, "ema", tol=1e-6)
    close(sh, sh_ref, BF16, "shadow", tol=1e-2)
    # against torch.optim.AdamW itself (train.py:176-181)
    q = torch.nn.Parameter(bufs[0].clone())
    opt = torch.optim.AdamW([q], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3)
    p2, m2, v2 = q.detach().clone(), torch.zeros(n, device=d), torch.zeros(n, device=d)
    for step in (1, 2):
        gr = rnd((n,), F32, 10 + step)
        q.grad = gr.clone()
        opt.step()
        ops.adamw_ema(p2, gr, m2, v2, None, None, n, 1e-3, 0.9, 0.999, 1e-8, 1e-3, step, 0.0, 1.0)
    close(p2, q.detach(), F32, "adamw vs torch.optim.AdamW", tol=1e-6)


def test_grad_scaler_and_fp16_shadow():
    """Device-resident dynamic loss scale (torch.cuda.amp.GradScaler's rule; Fabric "16-mixed", train.py:98) around the fused
    AdamW: unscaling, skip on inf/nan (EMA still moves), growth after `interval` clean steps, bias corrections that count
    only the steps taken -- against torch.optim.AdamW + torch's GradScaler arithmetic restated on the host."""
    d = dev()
    n = 4099  # not a multiple of 4: the check kernel's tail
    p = rnd((n,), F32, 1)
    m, v = torch.zeros(n, device=d), torch.zeros(n, device=d)
    ema = p.clone()
    sh = torch.empty(n, dtype=torch.float16, device=d)
    q = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([q], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3)
    ema_ref = p.clone()
    st = torch.zeros(4, device=d)
    ops.grad_scaler_init(st, 1024.0)
    scale, tracker, taken = 1024.0, 0, 0
    for step in range(1, 8):
        g = rnd((n,), F32, 10 + step)
        bad = step in (2, 5)
        gs = g * scale
        if bad:
            gs[n - 1 if step == 2 else 17] = float("inf") if step == 2 else float("nan")
        ops.grad_scaler_check(gs, n, st)
        assert st[2].item() == (1.0 if bad else 0.0)
        ops.adamw_ema(p, gs, m, v, ema, sh, n, 1e-3, 0.9, 0.999, 1e-8, 1e-3, step, 0.99, 1.0, scaler=st)
        ops.grad_scaler_update(st, 2.0, 0.5, 3)
        if bad:
            scale, tracker = scale * 0.5, 0
        else:
            q.grad = g.clone()
            opt.step()
            taken += 1
            tracker += 1
            if tracker == 3:
                scale, tracker = scale * 2.0, 0
        ema_ref.mul_(0.99).add_(q.detach(), alpha=0.01)
        assert st.tolist() == [scale, float(tracker), 0.0, float(taken)], (step, st.tolist())
        close(p, q.detach(), F32, f"scaled adamw step {step}", tol=2e-6)
        close(ema, ema_ref, F32, f"ema step {step}", tol=2e-6)
    assert torch.equal(sh, p.to(torch.float16))


@pytest.mark.parametrize("dt", [F32, BF16, F16])
@pytest.mark.parametrize("shape", [(2, 64, 512), (3, 16, 64), (1, 256, 128), (2, 4, 64), (5, 64, 128), (3, 64, 96), (2, 256, 512), (1, 192, 64), (3, 128, 512), (2, 512, 256)])  # 16-bit: T=64 and T=64n (forward: online softmax over key blocks; backward: 64x64 blocks, two kernels) on the matrix cores
def test_attention(shape, dt):
    B, T, C = shape
    d = dev()
    qkv = rnd((B * T, 3 * C), dt, 1, scale=1.5)
    o, o_ref = torch.empty((B * T, C), dtype=TD[dt], device=d), torch.empty((B * T, C), dtype=TD[dt], device=d)
    lse, lse_ref = torch.empty(B * T, device=d), torch.empty(B * T, device=d)
    ops.attention_forward(qkv, o, lse, B, T, C, dt)
    E.attention_forward(qkv, o_ref, lse_ref, B, T, C, dt)
    close(o, o_ref, dt, "attention fwd")
    close(lse, lse_ref, F32, "attention lse", tol=1e-4)
    do = rnd((B * T, C), dt, 2)
    dq, dq_ref = torch.empty_like(qkv), torch.empty_like(qkv)
    delta = torch.empty(B * T, device=d)
    ops.attention_backward(qkv, o_ref, do, lse_ref, delta, dq, B, T, C, dt)
    E.attention_backward(qkv, o_ref, do, lse_ref, delta, dq_ref, B, T, C, dt)
    close(dq, dq_ref, dt, "attention bwd")


def test_regenerated_noise_stream_and_the_kernels_that_consume_it():
    """The training step's eps = randn_like(x) (src/thor/pipelines.py:22-25) as a counter-based stream: c2w_philox_normal writes it,
    c2w_nchw_to_nhwc_noise / c2w_mse_loss_grad_noise regenerate it -- bit-identical to the launchers that read the written tensor;
    the stream itself is standard normal (moments, tails, seed dependence)."""
    d = dev()
    n = 4_000_003  # not a multiple of 4: the tail of the last block
    a, b, c = (torch.empty(n, device=d) for _ in range(3))
    ops.philox_normal(a, n, 1234567890123)
    ops.philox_normal(b, n, 1234567890123)
    ops.philox_normal(c, n, 1234567890124)
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.isfinite(a).all()
    assert abs(a.mean().item()) < 2e-3 and abs(a.var().item() - 1.0) < 3e-3
    assert abs((a**4).mean().item() - 3.0) < 3e-2 and abs((a**3).mean().item()) < 1e-2
    assert 5.0 < a.abs().max().item() < 6.5 and abs((a.abs() > 1.959964).float().mean().item() - 0.05) < 1e-3
    assert abs(torch.corrcoef(torch.stack((a[:-1], a[1:])))[0, 1].item()) < 2e-3  # neighbours (the two Box-Muller outputs) uncorrelated
    for dt in (F32, BF16, F16):
        for (B, C, HW, ldc) in [(3, 13, 32 * 32, 64), (2, 65, 64 * 64, 128), (2, 6, 15 * 15, 8)]:  # last: HW % 4 != 0, scalar loads
            seed = 77 + HW
            x = rnd((B, C, HW), F32, 1)
            eps = torch.empty(B * C * HW, device=d)
            ops.philox_normal(eps, eps.numel(), seed)
            musig = torch.rand(B, 2, device=d) + 0.1
            y0 = torch.full((B * HW, ldc), 7.0, dtype=TD[dt], device=d)
            y1 = y0.clone()
            ops.nchw_to_nhwc(x, eps, musig, y0, B, C, HW, ldc, dt)
            assert ops.nchw_to_nhwc_noise(x, seed, musig, y1, B, C, HW, ldc, dt)
            assert torch.equal(y0, y1)
            yy = rnd((B * HW, ldc), dt, 2)
            dy0, dy1 = torch.empty_like(yy), torch.empty_like(yy)
            l0, l1 = torch.zeros(1, device=d), torch.zeros(1, device=d)
            ops.mse_loss_grad(yy, eps, dy0, l0, B, C, HW, ldc, 0.37, dt)
            assert ops.mse_loss_grad_noise(yy, seed, dy1, l1, B, C, HW, ldc, 0.37, dt)
            assert torch.equal(dy0, dy1)
            assert abs(l0.item() - l1.item()) <= 1e-5 * abs(l0.item())  # atomics: summation order


def test_conv_random_shapes_product_vs_direct_kernel():
    """Seeded sweep over the shape space the dispatcher splits between its kernels (halo-patch 8x16 / 16x16 tiles, gather
    256- and 128-pixel tiles, parity-class TS2): the product path against the one-thread-per-output direct kernel (naive=1),
    both on the GPU, with bias / activation / residual / multiplier drawn at random."""
    rs = np.random.RandomState(1234)
    modes = [ops.CONV_S1, ops.CONV_S1, ops.CONV_S1, ops.CONV_S2, ops.CONV_UP, ops.CONV_TS2, ops.CONV_1X1]
    for it in range(48):
        mode = modes[rs.randint(len(modes))]
        dt = [F32, BF16][rs.randint(2)]
        if dt == BF16 and it % 2:
            dt = F16  # same byte layout, other matrix-core opcode and conversions
        B = int(rs.randint(1, 4))
        Hin, Win = int(rs.choice([4, 8, 12, 16, 24, 32, 40])), int(rs.choice([4, 8, 16, 32, 48]))
        if mode == ops.CONV_S2:
            Hin, Win = Hin * 2, Win * 2
        Cin = int(rs.choice([64, 128, 192])) if dt != F32 else int(rs.choice([32, 64, 96]))
        ldy = int(rs.choice([8, 64, 72, 128, 136, 200]))
        Cout = ldy if rs.rand() < 0.7 else max(8, ldy - 8 * int(rs.randint(0, 3)))
        wrows = Cout if rs.rand() < 0.7 else max(1, Cout - int(rs.randint(1, 8)))
        Hout, Wout = _out_hw(mode, Hin, Win)
        taps = 1 if mode == ops.CONV_1X1 else 9
        g = geom(B, Hin, Win, Cin, Hout, Wout, Cout, ldy, wrows, mode)
        x = rnd((B * Hin * Win, Cin), dt, 10 * it + 1)
        w = rnd((wrows, taps, Cin), dt, 10 * it + 2, scale=1.0 / math.sqrt(taps * Cin))
        bias = rnd((wrows,), F32, 10 * it + 3) if rs.rand() < 0.6 else None
        kw = {}
        if rs.rand() < 0.4:
            kw["act"] = ops.ACT_SILU
        if rs.rand() < 0.5:
            kw["res"] = rnd((B * Hout * Wout, ldy), dt, 10 * it + 4)
        if "act" not in kw and rs.rand() < 0.4:
            kw["mul"] = rnd((B * Hout * Wout, ldy), dt, 10 * it + 5)
            kw["mulmode"] = [ops.MUL_PLAIN, ops.MUL_DSILU][rs.randint(2)]
        y = torch.full((B * Hout * Wout, ldy), 7.0, dtype=TD[dt], device=dev())
        y_ref = y.clone()
        ops.conv(x, w, bias, y, g, dt, naive=0, **kw)
        ops.conv(x, w, bias, y_ref, g, dt, naive=1, **kw)
        torch.cuda.synchronize()
        close(y, y_ref, dt, f"random conv #{it}: mode={mode} dt={dt} B={B} {Hin}x{Win} {Cin}->{Cout}/{ldy} wrows={wrows} {sorted(kw)}",
              tol=2e-4 if dt == F32 else 2e-2)


def test_wgrad_random_shapes_vs_restatement():
    """Seeded sweep of the weight-gradient kernels (halo-patch / gather; atomics / workspace reduction) against autograd of the
    PyTorch restatement, including ragged channel counts and the bias gradient."""
    rs = np.random.RandomState(4321)
    modes = [ops.CONV_S1, ops.CONV_S1, ops.CONV_S2, ops.CONV_UP, ops.CONV_1X1]
    for it in range(24):
        mode = modes[rs.randint(len(modes))]
        dt = [F32, BF16][rs.randint(2)]
        if dt == BF16 and it % 2:
            dt = F16  # same byte layout, other matrix-core opcode and conversions
        ws = ops.new_workspace(dev()) if rs.rand() < 0.5 else None
        B = int(rs.randint(1, 4))
        Hin, Win = int(rs.choice([4, 8, 16, 24])), int(rs.choice([8, 16, 32]))
        if mode == ops.CONV_S2:
            Hin, Win = Hin * 2, Win * 2
        Cin = int(rs.choice([64, 128, 192])) if dt != F32 else int(rs.choice([32, 64, 96]))
        ldy = int(rs.choice([8, 64, 72, 128, 200]))
        Cw = ldy if rs.rand() < 0.6 else max(1, ldy - int(rs.randint(1, 9)))
        Hout, Wout = _out_hw(mode, Hin, Win)
        taps = 1 if mode == ops.CONV_1X1 else 9
        g = geom(B, Hin, Win, Cin, Hout, Wout, Cw, ldy, Cw, mode)
        x = rnd((B * Hin * Win, Cin), dt, 20 * it + 1)
        dy = rnd((B * Hout * Wout, ldy), dt, 20 * it + 2)
        dw = torch.zeros(Cw * taps * Cin, dtype=torch.float32, device=dev())
        db = torch.zeros(Cw, dtype=torch.float32, device=dev())
        dw_ref, db_ref = dw.clone(), db.clone()
        ops.conv_wgrad(x, dy, dw, g, dt, dbias=db, workspace=ws)
        E.conv_wgrad(x, dy, dw_ref, g, dt, dbias=db_ref)
        torch.cuda.synchronize()
        what = f"random wgrad #{it}: mode={mode} dt={dt} B={B} {Hin}x{Win} {Cin}->{Cw}/{ldy}"
        close(dw, dw_ref, dt, what, tol=1e-4 if dt == F32 else 1e-2)
        close(db, db_ref, dt, what + " bias", tol=1e-4 if dt == F32 else 1e-2)


@pytest.mark.parametrize("dt", [BF16, F16])
def test_big_kernels_are_run_to_run_identical_at_bench_size(dt):
    """The hand-counted vmcnt / lgkmcnt waits of conv_patch_t3_kernel and wgrad_patch_kernel leave no room for a race to hide in a
    single comparison: at the bench's own size (B = 128, 128 -> 128 @128x128, 8192 / 256 workgroups) forty launches each, alone and
    with the other kernel running beside them on a second stream, must reproduce the first result bit for bit (the split-K sums go
    through the workspace and are added in a fixed order), and that result must satisfy the size-independent property the domain
    offers: linearity in the weights / in dy."""
    B, H, C = 128, 128, 128
    g = geom(B, H, H, C, H, H, C, C, C, ops.CONV_S1)
    x = rnd((B * H * H, C), dt, 1)
    w = rnd((C, 9, C), dt, 2, scale=1.0 / math.sqrt(9 * C))
    bias = torch.randn(C, device=dev())
    dy = rnd((B * H * H, C), dt, 3)
    y0 = torch.empty_like(x)
    ops.conv(x, w, bias, y0, g, dt)
    ws = ops.new_workspace(dev())
    dw0 = torch.zeros(C * 9 * C, dtype=torch.float32, device=dev())
    ops.conv_wgrad(x, dy, dw0, g, dt, workspace=ws)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    ws2 = ops.new_workspace(dev())
    y, dw = torch.empty_like(y0), torch.empty_like(dw0)
    for it in range(40):
        concurrent = it % 2 == 1
        y.zero_()
        dw.zero_()
        torch.cuda.synchronize()
        if concurrent:
            with torch.cuda.stream(side):
                ops.conv_wgrad(x, dy, dw, g, dt, workspace=ws2)
            ops.conv(x, w, bias, y, g, dt)
        else:
            ops.conv(x, w, bias, y, g, dt)
            ops.conv_wgrad(x, dy, dw, g, dt, workspace=ws)
        torch.cuda.synchronize()
        assert torch.equal(y, y0), f"conv launch {it} (concurrent={concurrent}) differs from the first"
        assert torch.equal(dw, dw0), f"weight-gradient launch {it} (concurrent={concurrent}) differs from the first"
    # linearity: conv(x; 2w, 2b) = 2 conv(x; w, b) exactly in a binary floating-point format (normal range); dW(x, 2 dy) = 2 dW(x, dy)
    y2 = torch.empty_like(y0)
    ops.conv(x, (w.float() * 2).to(TD[dt]), bias * 2, y2, g, dt)
    dw2 = torch.zeros_like(dw0)
    ops.conv_wgrad(x, (dy.float() * 2).to(TD[dt]), dw2, g, dt, workspace=ws)
    torch.cuda.synchronize()
    if dt == BF16:
        assert torch.equal(y2.float(), y0.float() * 2)
        assert torch.equal(dw2, dw0 * 2)
    else:  # fp16: results below 2^-14 are subnormal and carry fewer bits than their doubles (measured: every mismatch is one of those)
        close(y2.float(), y0.float() * 2, dt, "conv linearity", tol=2e-3)
        close(dw2, dw0 * 2, dt, "weight-gradient linearity", tol=2e-3)


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("B,H,W,Cin,wrows,r0,nr", [
    (3, 16, 16, 64, 6, 2, 2),        # tiny net: F = 2, k = 1 (centre frame of three)
    (2, 32, 48, 128, 52, 24, 4),     # F = 4, k = 6: the reference recipe's centre frame; 4 x 3 tiles per image
    (5, 128, 128, 128, 65, 30, 5),   # F = 5, k = 6 at the benchmarked field size
    (2, 8, 16, 128, 20, 4, 16),      # sixteen kept rows, ending at the matrix's last row
    (1, 24, 32, 64, 6, 4, 2),        # the 16-row tile reaches past the matrix: rows >= wrows read as zeros
])
def test_conv_center_vs_emulation(dt, B, H, W, Cin, wrows, r0, nr):
    """c2w_conv_center (model/nn.py:194 restricted to what src/thor/score.py:76-88 keeps) against conv2d on the selected rows; the
    planes of other windows / channels in the destination stay untouched."""
    assert ops.conv_center_supported(H, W, Cin, nr, dt)
    x = rnd((B * H * W, Cin), dt, 1)
    w = rnd((wrows, 9, Cin), dt, 2, scale=1.0 / math.sqrt(9 * Cin))
    bias = torch.randn(wrows, generator=torch.Generator().manual_seed(3)).to(dev())
    nplanes = nr + 3
    ostride = nplanes * H * W
    out = torch.full((B, nplanes, H, W), -7.0, device=dev())
    ref = out.clone()
    ops.conv_center(x, w, bias, out, B, H, W, Cin, wrows, r0, nr, ostride, dt)
    E.conv_center(x.cpu(), w.cpu(), bias.cpu(), ref_cpu := ref.cpu(), B, H, W, Cin, wrows, r0, nr, ostride, dt)
    torch.cuda.synchronize()
    assert torch.equal(out[:, nr:].cpu(), ref_cpu[:, nr:])  # nothing written outside the kept planes
    close(out[:, :nr].cpu(), ref_cpu[:, :nr], dt, "conv_center")
    # the same rows out of the full convolution agree to the rounding of the compute type (different summation order)
    ldy = 128
    y = torch.zeros((B * H * W, ldy), dtype=TD[dt], device=dev())
    ops.conv(x, w, bias, y, geom(B, H, W, Cin, H, W, min(128, ((wrows + 15) // 16) * 16), ldy, wrows, ops.CONV_S1), dt)
    full = y.float().view(B, H * W, ldy)[:, :, r0:r0 + nr].permute(0, 2, 1).reshape(B, nr, H, W)
    close(out[:, :nr], full, dt, "conv_center vs full conv", tol=2 * TOL[dt] if dt == BF16 else None)
    assert not ops.conv_center_supported(H + 4, W, Cin, nr, dt) and not ops.conv_center_supported(H, W, 192, nr, dt)
    assert not ops.conv_center_supported(H, W, Cin, 17, dt) and not ops.conv_center_supported(H, W, Cin, nr, F32)


@pytest.mark.parametrize("rows,K,ldk,act", [(512, 32, 32, ops.ACT_SILU), (512, 512, 512, ops.ACT_SILU), (8448, 512, 512, ops.ACT_NONE),
                                            (64, 5, 32, ops.ACT_NONE), (7, 33, 33, ops.ACT_RELU)])
def test_gemv_f32_vs_emulation(rows, K, ldk, act):
    """c2w_gemv_f32: the time-embedding MLP / modulation projections for ONE t (model/score.py:56-57,62-67; model/nn.py:149); also
    against the fp32 matrix-core GEMM the batched path uses for the same layer."""
    g = torch.Generator().manual_seed(rows + K)
    x = torch.randn(ldk, generator=g).to(dev())
    Wm = (torch.randn(rows, ldk, generator=g) / math.sqrt(K)).to(dev())
    b = torch.randn(rows, generator=g).to(dev())
    y = torch.full((rows + 3,), -7.0, device=dev())
    ops.gemv_f32(x, Wm, b, y, rows, K, ldk, act)
    ref = torch.full((rows + 3,), -7.0)
    E.gemv_f32(x.cpu(), Wm.cpu(), b.cpu(), ref, rows, K, ldk, act)
    torch.cuda.synchronize()
    assert torch.equal(y[rows:].cpu(), ref[rows:])
    close(y[:rows].cpu(), ref[:rows], F32, "gemv")
    if ldk % 32 == 0 and K == ldk and act != ops.ACT_RELU:
        y2 = torch.zeros((1, rows), device=dev())
        ops.conv(x.view(1, ldk), Wm, b, y2, geom(1, 1, 1, ldk, 1, 1, rows, rows, rows, ops.CONV_1X1), F32, act=act)
        close(y[:rows], y2.view(-1), F32, "gemv vs fp32 MFMA linear")
