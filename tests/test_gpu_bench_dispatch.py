"""Parity at the dispatch the bench times.

bench.py runs the default network at B = 128, C = 65 in bf16; at that batch the dispatcher (c2w_conv_dispatch) sends the 128^2, 64^2
and 32^2 levels, three of the four up-convs and the padded edge convs to conv_patch_t3_kernel<16> (>= 512 workgroups since round 6; 1024 before), the weight
gradients to wgrad_patch_kernel with one-round split counts, the 8x8 level to the paired tiles.  The per-kernel tests of
test_gpu_kernels.py stay small (seconds on any box) and therefore reach the 16x16-tile kernel at one shape only; this file covers

  (a) every (kernel family, geometry, epilogue) combination of THAT step -- model/nn.py:146-159 (residual block), :165-194 (head /
      down / up / out convs), :31-85 (attention block) -- against the plain PyTorch restatement (tests/emu_ops.py) on the same seeded
      inputs, each case asserting the kernel family it reaches;
  (b) Trainer's own forward + backward (training_loop.py:376-378) at B = 128, C = 65 in bf16 and fp16 against the CPU oracle's
      gradients computed on the box in chunks: loss, all 228 gradients.

Tolerances: as in test_gpu_kernels.py (of the output scale): bf16 2e-2, fp16 4e-3, weight gradients 1e-2 / 2e-3.
"""
import math
import os

import pytest
import torch

import emu_ops as E
from climate2weather_amd import _lib, ops

pytestmark = pytest.mark.gpu

BF16, F16, F32 = ops.DTYPE_BF16, ops.DTYPE_F16, ops.DTYPE_F32
TD = ops.TORCH_DTYPE
TOL = {BF16: 2e-2, F16: 4e-3}
TOL_W = {BF16: 1e-2, F16: 2e-3}
S1, S2, UP, TS2, K1 = ops.CONV_S1, ops.CONV_S2, ops.CONV_UP, ops.CONV_TS2, ops.CONV_1X1
T3, HALF, PAIR, TS2P, S2P, GATHER = _lib.KERNEL_PATCH_16X16, _lib.KERNEL_PATCH_8X16, _lib.KERNEL_PATCH_PAIR, _lib.KERNEL_PATCH_TS2, _lib.KERNEL_PATCH_S2, _lib.KERNEL_GATHER


def dev():
    return torch.device("cuda:0")


def rnd(shape, dt, seed, scale=1.0):
    g = torch.Generator(device=dev()).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=dev()) * scale).to(TD[dt])


def close(a, b, tol, what):
    a, b = a.float(), b.float()
    s = max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item()
    assert math.isfinite(err) and err <= tol * s, f"{what}: max err {err:.3e} vs scale {s:.3e}"


def geom(B, Hin, Win, Cin, Hout, Wout, Cout, ldy, wrows, mode):
    return dict(B=B, Hin=Hin, Win=Win, Cin=Cin, Hout=Hout, Wout=Wout, Cout=Cout, ldy=ldy, wrows=wrows, mode=mode)


def out_hw(mode, H):
    return H // 2 if mode == S2 else 2 * H if mode in (UP, TS2) else H


# (name, kernel family, mode, B, Hin, Cin, Cout, wrows, cin_real, epilogues)
# Epilogue names: bias, silu (inference), pair (training: silu + silu'), res, lnf (consumer's LayerNorm emitted; "mod" = with the
# modulation rows), lnb (LayerNorm backward fused), mulp (x stored silu' -- conv2's input gradient), pool (2x2 sums: the up-conv's
# input gradient).  B is the smallest batch at which the launch has the bench's kernel selection (>= 512 workgroups, 1024 in rounds 1-5, for the 16x16
# tiles; for the gather kernel the 256-pixel-tile form).
FWD_CASES = [
    # residual blocks (model/nn.py:146-159): conv1 / conv2 / their input gradients
    ("res 128@64^2", T3, S1, 64, 64, 128, 128, 128, 128, ["bias+pair", "bias+res+lnf", "mulp", "res+lnb", "bias+silu"]),
    ("res 256@32^2 (two co tiles, 4 K chunks)", T3, S1, 128, 32, 256, 256, 256, 256, ["bias+pair", "bias+res", "mulp", "plain", "bias+silu"]),
    ("res 384@16^2", HALF, S1, 128, 16, 384, 384, 384, 384, ["bias+pair", "bias+res", "mulp", "plain"]),
    ("res 512@8^2 (paired images)", PAIR, S1, 128, 8, 512, 512, 512, 512, ["bias+pair", "bias+res", "mulp", "plain"]),
    # up-convs (model/nn.py:183-189) with the skip add (:238) and, where the consumer is a 128-channel block, its LayerNorm
    ("up 512->384 8^2->16^2", HALF, UP, 128, 8, 512, 384, 384, 512, ["bias+res"]),
    ("up 384->256 16^2->32^2", T3, UP, 128, 16, 384, 256, 256, 384, ["bias+res"]),
    ("up 256->128 32^2->64^2", T3, UP, 64, 32, 256, 128, 128, 256, ["bias+res", "bias+res+lnf"]),
    # input gradients of the up-convs, leaving the kernel as 2x2 sums (adjoint of Upsample, model/nn.py:184)
    ("up-conv dgrad 256->384 @32^2 pooled (three co tiles)", T3, S1, 128, 32, 256, 384, 384, 256, ["pool"]),
    ("up-conv dgrad 128->256 @64^2 pooled", T3, S1, 32, 64, 128, 256, 256, 128, ["pool"]),
    ("up-conv dgrad 384->512 @16^2 pooled (512 workgroups: the 16x16-tile kernel since round 6)", T3, S1, 128, 16, 384, 512, 512, 384, ["pool"]),
    ("up-conv dgrad 384->512 @16^2 pooled at half the batch (8x16 tiles)", HALF, S1, 64, 16, 384, 512, 512, 384, ["pool"]),
    # edge convs (model/nn.py:193-194) at C = 65: K padded 65 -> 128 with zero channels; 65 real output rows in a 128-wide buffer
    ("network input 65(128)->128 @128^2", T3, S1, 16, 128, 128, 128, 128, 65, ["bias+lnf", "bias"]),
    ("network output 128->65(128) @128^2", T3, S1, 16, 128, 128, 128, 65, 128, ["bias"]),
    ("network output's input gradient 65(128)->128 @128^2", T3, S1, 16, 128, 128, 128, 128, 65, ["plain"]),
    ("K of 3 chunks, the last one half void: 160(192)->128 @64^2", T3, S1, 64, 64, 192, 128, 128, 160, ["bias+res"]),
    ("K promise that leaves whole chunks out: 40(192)->128 @64^2", T3, S1, 64, 64, 192, 128, 128, 40, ["bias"]),
    # stride-2 family (model/nn.py:169-174): forward on the gather kernel (the parity-plane halo kernel of round 6 is opt-in:
    # S2_FWD_CASES below), input gradient per output-parity class (+ skip gradient)
    ("down 128->128 128^2->64^2", GATHER, S2, 16, 128, 128, 128, 128, 128, ["bias"]),
    ("down 256->384 32^2->16^2", GATHER, S2, 128, 32, 256, 384, 384, 256, ["bias"]),
    ("down dgrad 128->128 64^2->128^2", TS2P, TS2, 16, 64, 128, 128, 128, 128, ["res"]),
    ("down dgrad 384->256 16^2->32^2", TS2P, TS2, 128, 16, 384, 256, 256, 384, ["res"]),
    ("down dgrad 512->384 8^2->16^2", GATHER, TS2, 128, 8, 512, 384, 384, 512, ["res"]),
]


def _run_conv_case(case, dt, ep):
    name, fam, mode, B, Hin, Cin, Cout, wrows, cin_real, _ = case
    Hout = out_hw(mode, Hin)
    g = geom(B, Hin, Hin, Cin, Hout, Hout, Cout, Cout, wrows, mode)
    parts = ep.split("+")
    pool = "pool" in parts
    want_lnf, want_lnb = "lnf" in parts, "lnb" in parts
    assert ops.conv_dispatch(g, dt, pool2=pool, fused_ln=want_lnf or want_lnb) == fam, f"{name}: not on the kernel family this case is for"
    npix_in, npix = B * Hin * Hin, B * Hout * Hout
    x = rnd((npix_in, Cin), dt, 1)
    w = rnd((wrows, 9, Cin), dt, 2, scale=1.0 / math.sqrt(9 * cin_real))
    if cin_real < Cin:  # the padded network-input operand: channels >= 65 are zero in both operands (engine._w, nchw_to_nhwc)
        x[:, cin_real:] = 0
        w[:, :, cin_real:] = 0
    bias = rnd((wrows,), F32, 3) if "bias" in parts else None
    res = rnd((npix, Cout), dt, 4) if "res" in parts else None
    m = rnd((B, Cout + 64), F32, 6)
    rows_out = npix // 4 if pool else npix
    y = torch.full((rows_out, Cout), 7.0, dtype=TD[dt], device=dev())
    y_ref = y.clone()
    kw, kw_ref, extra = {}, {}, []
    if "silu" in parts:
        kw["act"] = ops.ACT_SILU
    if "pair" in parts:
        y2, y2_ref = torch.full_like(y, 3.0), torch.full_like(y, 3.0)
        kw, kw_ref = dict(act=ops.ACT_SILU_PAIR, y2=y2), dict(act=ops.ACT_SILU_PAIR, y2=y2_ref)
        extra.append((y2, y2_ref, "silu' output"))
    if "mulp" in parts:
        kw["mul"] = rnd((npix, Cout), dt, 5)
        kw["mulmode"] = ops.MUL_PLAIN
    if pool:
        kw["pool2"] = True
    if want_lnf:
        assert ops.conv_lnfwd_supported(g, dt)
        hn, hn_ref = torch.full_like(y, 3.0), torch.full_like(y, 3.0)
        lnf = dict(m=m.view(-1)[32:], ldm=Cout + 64, eps=1e-5, unbiased=True)
        kw, kw_ref = dict(lnf=dict(lnf, y=hn)), dict(lnf=dict(lnf, y=hn_ref))
        extra.append((hn, hn_ref, "fused LayerNorm output"))
    if want_lnb:
        assert ops.conv_lnbwd_supported(g, dt)
        dm, dm_ref = torch.zeros_like(m), torch.zeros_like(m)
        ln = dict(x=rnd((npix, Cout), dt, 7), m=m.view(-1)[32:], ldm=Cout + 64, eps=1e-5, unbiased=True)
        kw, kw_ref = dict(ln=dict(ln, dm=dm.view(-1)[32:])), dict(ln=dict(ln, dm=dm_ref.view(-1)[32:]))
    if not kw_ref:
        kw_ref = kw
    ops.conv(x, w, bias, y, g, dt, res=res, **kw)
    E.conv(x, w, bias, y_ref, g, dt, res=res, **kw_ref)
    torch.cuda.synchronize()
    close(y, y_ref, TOL[dt], f"{name} [{ep}]")
    for a, b, what in extra:
        close(a, b, TOL[dt], f"{name} [{ep}] {what}")
    if want_lnb:
        close(dm, dm_ref, 1e-2, f"{name} [{ep}] modulation gradient")
    if wrows < Cout:
        assert y[:, wrows:].abs().max().item() == 0.0, f"{name}: padded output rows must stay zero"
    if cin_real < Cin:  # C2wConvArgs.kvalid: the promise that the padding channels are zero lets the kernel skip K steps -- same bits
        y_k = torch.full_like(y, 5.0)
        kw_k = dict(kw)
        if want_lnf:
            hn_k = torch.full_like(y, 5.0)
            kw_k = dict(lnf=dict(kw["lnf"], y=hn_k))
        ops.conv(x, w, bias, y_k, g, dt, res=res, kvalid=cin_real, **kw_k)
        assert torch.equal(y_k, y), f"{name} [{ep}]: kvalid = {cin_real} changed the result"
        if want_lnf:
            assert torch.equal(hn_k, hn)


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("case", FWD_CASES, ids=[c[0] for c in FWD_CASES])
def test_conv_kernels_at_the_bench_dispatch(case, dt):
    for ep in case[-1]:
        _run_conv_case(case, dt, ep)


# the four down-convs at the bench's batch on the parity-plane halo kernel (C2W_CONV_S2_PATCH=1: opt-in, see _lib.HOST_KNOB_DEFAULTS)
S2_FWD_CASES = [
    ("down 128->128 128^2->64^2", S2P, S2, 16, 128, 128, 128, 128, 128, ["bias"]),
    ("down 128->256 64^2->32^2 (two output-channel tiles)", S2P, S2, 128, 64, 128, 256, 256, 128, ["bias", "bias+silu", "bias+res"]),
    ("down 256->384 32^2->16^2", S2P, S2, 128, 32, 256, 384, 384, 256, ["bias", "plain"]),
    ("down 384->512 16^2->8^2 (8-pixel-wide output: two images per tile)", S2P, S2, 128, 16, 384, 512, 512, 384, ["bias", "bias+silu"]),
]


@pytest.fixture
def s2_patch_on(monkeypatch):
    monkeypatch.setenv("C2W_CONV_S2_PATCH", "1")
    ops.knobs_reload()
    yield
    monkeypatch.delenv("C2W_CONV_S2_PATCH")
    ops.knobs_reload()  # back to the host default (off)


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("case", S2_FWD_CASES, ids=[c[0] for c in S2_FWD_CASES])
def test_stride2_forward_kernel_at_the_bench_dispatch(case, dt, s2_patch_on):
    for ep in case[-1]:
        _run_conv_case(case, dt, ep)


def test_stride2_forward_dispatch_rule(monkeypatch):
    """Off unless C2W_CONV_S2_PATCH says otherwise (host default); with =1 the parity-plane kernel is taken from four K chunks on or up to 2048 workgroups (one workgroup per CU: short chains in big launches
    lose to the gather kernel, profiles/r06x_ab_s2_forward.txt); C2W_CONV_S2_PATCH=2 takes it wherever the geometry allows, =0 never."""
    big = geom(128, 128, 128, 128, 64, 64, 128, 128, 128, S2)    # 4096 workgroups, two K chunks
    deep = geom(128, 32, 32, 256, 16, 16, 384, 384, 256, S2)     # four K chunks
    mid = geom(128, 64, 64, 128, 32, 32, 256, 256, 128, S2)      # 2048 workgroups
    narrow = geom(128, 16, 16, 384, 8, 8, 512, 512, 384, S2)     # 8-pixel-wide output: two images per tile, six K chunks
    odd = geom(128, 24, 24, 128, 12, 12, 128, 128, 128, S2)      # 12-pixel-wide output: no tiling covers it
    assert [ops.conv_dispatch(g, BF16) for g in (big, deep, mid, narrow, odd)] == [GATHER] * 5  # the host's default: off (_lib.HOST_KNOB_DEFAULTS)
    try:
        monkeypatch.setenv("C2W_CONV_S2_PATCH", "1")
        ops.knobs_reload()
        assert [ops.conv_dispatch(g, BF16) for g in (big, deep, mid, narrow, odd)] == [GATHER, S2P, S2P, S2P, GATHER]
        assert ops.conv_dispatch(deep, F32) == GATHER
        monkeypatch.setenv("C2W_CONV_S2_PATCH", "2")
        ops.knobs_reload()
        assert [ops.conv_dispatch(g, F16) for g in (big, deep, mid, narrow, odd)] == [S2P, S2P, S2P, S2P, GATHER]
    finally:
        monkeypatch.delenv("C2W_CONV_S2_PATCH")
        ops.knobs_reload()
    assert ops.conv_dispatch(deep, BF16) == GATHER


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("case", [("one tile, one K chunk, one image", 1, 16, 32, 64, 128),
                                  ("non-square 32x64 -> 16x32 (four tiles), three K chunks, three images", 3, 32, 64, 192, 128),
                                  ("192 output channels (the second channel tile half empty), two K chunks", 2, 32, 32, 128, 192),
                                  ("48 x 96 -> 24 x 48: tile rows and columns in the image's interior and at all four edges", 2, 48, 96, 64, 128),
                                  ("8-pixel-wide output, THREE images (the last tile's partner is missing), two K chunks", 3, 16, 16, 128, 128),
                                  ("8-pixel-wide output of 16 rows (two tiles per image pair), four images, 192 output channels", 4, 32, 16, 64, 192)],
                         ids=["tile", "nonsquare", "cout192", "edges", "pair-odd", "pair-tall"])
def test_stride2_forward_on_parity_planes_against_the_gather_kernel(case, dt, monkeypatch, s2_patch_on):
    """Round 6, conv_patch_s2_kernel (model/nn.py:169-174 forward): the parity-plane halo kernel against the PyTorch restatement AND
    against the gather kernel it replaces (C2W_CONV_S2_PATCH=0) on shapes that exercise what the bench's launches do not: a single tile
    (every patch edge is padding), non-square images, a partial output-channel tile, several K chunks (the two row parities refilled in
    turn), 8-pixel-wide outputs (two images per tile; an odd batch), bias / activation / residual epilogues."""
    name, B, Hin, Win, Cin, Cout = case
    Hout, Wout = Hin // 2, Win // 2
    g = geom(B, Hin, Win, Cin, Hout, Wout, Cout, Cout, Cout, S2)
    assert ops.conv_dispatch(g, dt) == S2P, name
    x = rnd((B * Hin * Win, Cin), dt, 1)
    w = rnd((Cout, 9, Cin), dt, 2, scale=1.0 / math.sqrt(9 * Cin))
    bias = rnd((Cout,), F32, 3)
    res = rnd((B * Hout * Wout, Cout), dt, 4)
    for kw in (dict(), dict(act=ops.ACT_SILU), dict(res=res)):
        y = torch.full((B * Hout * Wout, Cout), 7.0, dtype=TD[dt], device=dev())
        y_ref, y_g = y.clone(), y.clone()
        ops.conv(x, w, bias, y, g, dt, **kw)
        E.conv(x, w, bias, y_ref, g, dt, **kw)
        monkeypatch.setenv("C2W_CONV_S2_PATCH", "0")
        ops.knobs_reload()
        assert ops.conv_dispatch(g, dt) == GATHER
        ops.conv(x, w, bias, y_g, g, dt, **kw)
        monkeypatch.setenv("C2W_CONV_S2_PATCH", "1")
        ops.knobs_reload()
        torch.cuda.synchronize()
        close(y, y_ref, TOL[dt], f"{name} {sorted(kw)}")
        close(y, y_g, TOL[dt], f"{name} {sorted(kw)} against the gather kernel")


# (name, kernel family, mode, B, Hin, Cin, rows, ldy, cin_real)
WGRAD_CASES = [
    ("128->128 @128^2, B = 128 (the dominant weight gradient at the bench's split count)", HALF, S1, 128, 128, 128, 128, 128, 128),
    ("128->128 @64^2", HALF, S1, 128, 64, 128, 128, 128, 128),
    ("256->256 @32^2", HALF, S1, 128, 32, 256, 256, 256, 256),
    ("384->384 @16^2", HALF, S1, 128, 16, 384, 384, 384, 384),
    ("512->512 @8^2 (paired images)", PAIR, S1, 128, 8, 512, 512, 512, 512),
    ("up 512->384 8^2->16^2", HALF, UP, 128, 8, 512, 384, 384, 512),
    ("up 384->256 16^2->32^2", HALF, UP, 128, 16, 384, 256, 256, 384),
    ("up 256->128 32^2->64^2", HALF, UP, 128, 32, 256, 128, 128, 256),
    ("up 128->128 64^2->128^2", HALF, UP, 32, 64, 128, 128, 128, 128),
    ("network input 65(128)->128 @128^2", HALF, S1, 32, 128, 128, 128, 128, 65),
    ("network output 128->65 @128^2 (65 gradient rows of a 128-wide dY)", HALF, S1, 32, 128, 128, 65, 128, 128),
    ("down 128->128 128^2->64^2", GATHER, S2, 32, 128, 128, 128, 128, 128),
    ("down 384->512 16^2->8^2", GATHER, S2, 128, 16, 384, 512, 512, 384),
]


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("case", WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_weight_gradient_kernels_at_the_bench_dispatch(case, dt):
    """c2w_conv_wgrad with the per-call workspace (the engine's path: split-K partial sums stored, then reduced in a fixed order) on the
    geometries and batch of the bench's step, against autograd of the PyTorch restatement; a second run must reproduce the first bit
    for bit."""
    name, fam, mode, B, Hin, Cin, rows, ldy, cin_real = case
    Hout = out_hw(mode, Hin)
    g = geom(B, Hin, Hin, Cin, Hout, Hout, rows, ldy, rows, mode)
    assert ops.conv_wgrad_dispatch(g, dt) == fam, name
    x = rnd((B * Hin * Hin, Cin), dt, 1)
    if cin_real < Cin:
        x[:, cin_real:] = 0
    dy = rnd((B * Hout * Hout, ldy), dt, 2)
    if rows < ldy:
        dy[:, rows:] = 0  # what mse_loss_grad leaves in the padding channels
    ws = ops.new_workspace(dev())
    need = ops.conv_wgrad_workspace_bytes(g, dt)
    assert 0 < need <= ops.WORKSPACE_BYTES, f"{name}: expected a split reduction through the workspace"
    dw = torch.zeros(rows * 9 * Cin + 64, dtype=torch.float32, device=dev())
    db = torch.zeros(rows + 8, dtype=torch.float32, device=dev())
    dw_ref, db_ref = dw.clone(), db.clone()
    ops.conv_wgrad(x, dy, dw, g, dt, dbias=db, workspace=ws)
    E.conv_wgrad(x, dy, dw_ref, g, dt, dbias=db_ref)
    torch.cuda.synchronize()
    close(dw, dw_ref, TOL_W[dt], name)
    close(db, db_ref, TOL_W[dt], name + " bias")
    assert dw[-64:].abs().max().item() == 0.0 and db[rows:].abs().max().item() == 0.0
    if cin_real < Cin:
        assert dw[: rows * 9 * Cin].view(rows, 9, Cin)[:, :, cin_real:].abs().max().item() == 0.0
    dw2, db2 = torch.zeros_like(dw), torch.zeros_like(db)
    ops.conv_wgrad(x, dy, dw2, g, dt, dbias=db2, workspace=ws)
    assert torch.equal(dw, dw2), f"{name}: the workspace reduction must be deterministic"


@pytest.mark.parametrize("dt", [BF16, F16])
def test_attention_block_kernels_at_the_bench_batch(dt):
    """model/nn.py:31-85 at B = 128, 8x8 tokens, 512 channels: LayerNorm over the channel axis, qkv / proj 1x1 convs and their input /
    weight gradients (gather kernels, 8192 "pixels"), the matrix-core attention forward and backward."""
    B, T, C = 128, 64, 512
    npix = B * T
    x = rnd((npix, C), dt, 1)
    hl, hl_ref = torch.empty_like(x), torch.empty_like(x)
    ops.ln_forward(x, None, hl, npix, T, C, 0, 1e-5, True, dt)
    E.ln_forward(x, None, hl_ref, npix, T, C, 0, 1e-5, True, dt)
    close(hl, hl_ref, TOL[dt], "LayerNorm(512) forward")
    for rows, use_res in ((3 * C, False), (C, True)):
        g = geom(npix, 1, 1, C, 1, 1, rows, rows, rows, K1)
        assert ops.conv_dispatch(g, dt) == GATHER
        w = rnd((rows, 1, C), dt, 2, scale=1.0 / math.sqrt(C))
        bias = rnd((rows,), F32, 3)
        res = rnd((npix, rows), dt, 4) if use_res else None
        y = torch.full((npix, rows), 7.0, dtype=TD[dt], device=dev())
        y_ref = y.clone()
        ops.conv(hl_ref, w, bias, y, g, dt, res=res)
        E.conv(hl_ref, w, bias, y_ref, g, dt, res=res)
        close(y, y_ref, TOL[dt], f"1x1 512->{rows}")
        dy = rnd((npix, rows), dt, 5)
        dw = torch.zeros(rows * C, dtype=torch.float32, device=dev())
        db = torch.zeros(rows, dtype=torch.float32, device=dev())
        dw_ref, db_ref = dw.clone(), db.clone()
        ops.conv_wgrad(hl_ref, dy, dw, g, dt, dbias=db, workspace=ops.new_workspace(dev()))
        E.conv_wgrad(hl_ref, dy, dw_ref, g, dt, dbias=db_ref)
        close(dw, dw_ref, TOL_W[dt], f"1x1 512->{rows} weight gradient")
        close(db, db_ref, TOL_W[dt], f"1x1 512->{rows} bias gradient")
        gd = geom(npix, 1, 1, rows, 1, 1, C, C, C, K1)  # input gradient: the same GEMM over dy with the transposed weights
        wT = rnd((C, 1, rows), dt, 6, scale=1.0 / math.sqrt(rows))
        dx, dx_ref = torch.empty((npix, C), dtype=TD[dt], device=dev()), torch.empty((npix, C), dtype=TD[dt], device=dev())
        ops.conv(dy, wT, None, dx, gd, dt)
        E.conv(dy, wT, None, dx_ref, gd, dt)
        close(dx, dx_ref, TOL[dt], f"1x1 {rows}->512 input gradient")
    qkv = rnd((npix, 3 * C), dt, 7, scale=1.5)
    o, o_ref = torch.empty((npix, C), dtype=TD[dt], device=dev()), torch.empty((npix, C), dtype=TD[dt], device=dev())
    lse, lse_ref = torch.empty(npix, device=dev()), torch.empty(npix, device=dev())
    ops.attention_forward(qkv, o, lse, B, T, C, dt)
    E.attention_forward(qkv, o_ref, lse_ref, B, T, C, dt)
    close(o, o_ref, TOL[dt], "attention forward")
    close(lse, lse_ref, 1e-4, "attention lse")
    do = rnd((npix, C), dt, 8)
    dq, dq_ref = torch.empty_like(qkv), torch.empty_like(qkv)
    delta = torch.empty(npix, device=dev())
    ops.attention_backward(qkv, o_ref, do, lse_ref, delta, dq, B, T, C, dt)
    E.attention_backward(qkv, o_ref, do, lse_ref, delta, dq_ref, B, T, C, dt)
    close(dq, dq_ref, TOL[dt], "attention backward")


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("HW,C", [(1024, 256), (256, 384), (64, 512), (4096, 128)])
def test_layernorm_passes_at_the_bench_batch(HW, C, dt):
    """The separate LayerNorm passes of the step (the 256 / 384 / 512-channel levels, where the conv epilogues do not hold a whole
    channel row; 128 channels for the up-block's input) at B = 128 with per-sample modulation rows."""
    B = 128
    npix = B * HW
    x = rnd((npix, C), dt, 1)
    m = rnd((B, C + 64), F32, 2)
    mm, ldm = m.view(-1)[32:], C + 64
    y, y_ref = torch.empty_like(x), torch.empty_like(x)
    ops.ln_forward(x, mm, y, npix, HW, C, ldm, 1e-5, True, dt)
    E.ln_forward(x, mm, y_ref, npix, HW, C, ldm, 1e-5, True, dt)
    close(y, y_ref, TOL[dt], "ln fwd")
    dy, dres = rnd((npix, C), dt, 3), rnd((npix, C), dt, 4)
    dx, dx_ref = torch.empty_like(x), torch.empty_like(x)
    dm, dm_ref = torch.zeros_like(m), torch.zeros_like(m)
    ops.ln_backward(dy, x, mm, dres, dx, dm.view(-1)[32:], npix, HW, C, ldm, 1e-5, True, dt)
    E.ln_backward(dy, x, mm, dres, dx_ref, dm_ref.view(-1)[32:], npix, HW, C, ldm, 1e-5, True, dt)
    close(dx, dx_ref, TOL[dt], "ln bwd dx")
    close(dm, dm_ref, 1e-2, "ln bwd dm")


# ----------------------------------------------------------------------------------------------------------------------------------
DEFAULT = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3,
               padding_mode="zeros", attention_levels=[4])
# (loss, per-tensor |g|_2, full gradient tensors) of the scale; observed values are appended to gpurun_out/bench_step_parity.txt
STEP_TOL = {"bf16": (3e-4, 1e-2, 4e-2), "fp16": (3e-5, 2e-3, 8e-3)}
# round 6: (relative L2 |a - b|_2 / |b|_2 <=, cosine >=) of EVERY gradient tensor in full (all 228) -- errors concentrated in a tensor's
# small entries do not hide behind its largest one; about twice the observed worst case (profiles/r06_bench_step_parity.txt)
# observed at B = 128 (r06 first GPU call): bf16 8.7e-3 / 1 - 3.8e-5, fp16 1.13e-3 / 1 - 6.3e-7
STEP_L2COS = {"bf16": (2e-2, 1.0 - 1e-4), "fp16": (3e-3, 1.0 - 2e-6)}


def test_trainer_forward_backward_at_bench_size_vs_cpu_oracle():
    """The timed path itself against the oracle: Trainer._forward_backward (what bench.py's step runs before the optimizer:
    training_loop.py:376-378) on the default network at the bench's own B = 128, C = 65, 128x128, in bf16 and in fp16, with injected
    (t, eps); the CPU oracle (oracle/unet.py, oracle/diffusion.py) computes loss and all 228 gradients of the same batch in 8 chunks
    of 16 windows on the box's host cores (about a minute).  Loss, every gradient's L2 norm and every gradient tensor in full."""
    from climate2weather_amd.score import ScoreUNet
    from climate2weather_amd.training import Trainer
    from oracle import diffusion as od
    from oracle import unet as ou
    B, C, H, CH = 128, 65, 128, 16
    torch.manual_seed(0)
    net = ScoreUNet(channels=C, spatial=2, activation=torch.nn.SiLU, **DEFAULT)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    gen = torch.Generator().manual_seed(128)
    x = torch.randn(B, C, H, H, generator=gen) * 0.5 + 0.5
    t = torch.rand(B, generator=gen)
    eps = torch.randn(B, C, H, H, generator=gen)
    # ---- oracle, chunked: d/dtheta of sum_chunk((y - eps)^2) / N accumulates to the gradient of the batch mean
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))  # BASELINE.md section 3 sweep: more threads are slower on the 256-thread hosts
    N = float(B * C * H * H)
    names = list(sd)
    g_ref = [torch.zeros_like(v) for v in sd.values()]
    loss_ref = 0.0
    for i in range(0, B, CH):
        xt = od.perturb(x[i:i + CH], t[i:i + CH].view(-1, 1, 1, 1), eps[i:i + CH])
        y = ou.score_unet_forward(sd, xt, t[i:i + CH], DEFAULT["hidden_blocks"], DEFAULT["attention_levels"])
        part = ((y - eps[i:i + CH]) ** 2).sum() / N
        for acc, gr in zip(g_ref, torch.autograd.grad(part, list(sd.values()))):
            acc += gr
        loss_ref += part.item()
        del y, part, xt
    torch.set_num_threads(nthreads)
    # ---- the product path
    net = net.cuda()
    xd, td, epsd = x.cuda(), t.cuda(), eps.cuda()
    report = []
    for mode, (tl, tn, tg) in STEP_TOL.items():
        tr = Trainer(net, precision=mode, ema_rates=())
        tr.eng.flat_grad.zero_()
        loss = tr._forward_backward(xd, td, epsd, sync=False)
        torch.cuda.synchronize()
        S = tr.loss_scale()
        e_loss = abs(loss.item() - loss_ref) / loss_ref
        named = dict(net.named_parameters())
        e_norm, e_full, e_l2, e_cos = (0.0, ""), (0.0, ""), (0.0, ""), (2.0, "")
        for n, gr in zip(names, g_ref):
            got = named[n].grad.detach().double().cpu() / S
            ref = gr.double()
            e_norm = max(e_norm, (abs(got.norm().item() - ref.norm().item()) / ref.norm().item(), n))
            e_full = max(e_full, ((got - ref).abs().max().item() / ref.abs().max().item(), n))
            e_l2 = max(e_l2, ((got - ref).norm().item() / ref.norm().item(), n))
            e_cos = min(e_cos, ((got.flatten() @ ref.flatten()).item() / (got.norm().item() * ref.norm().item()), n))
        report.append(f"B={B} C={C} {mode}: loss {e_loss:.2e}  |g|_2 {e_norm[0]:.2e} ({e_norm[1]})  full tensors {e_full[0]:.2e} ({e_full[1]})  "
                      f"| all 228 tensors: rel-L2 {e_l2[0]:.2e} ({e_l2[1]})  1-cos {1.0 - e_cos[0]:.2e} ({e_cos[1]})")
        assert e_loss <= tl, report[-1]
        assert e_norm[0] <= tn, report[-1]
        assert e_full[0] <= tg, report[-1]
        assert e_l2[0] <= STEP_L2COS[mode][0] and e_cos[0] >= STEP_L2COS[mode][1], report[-1]
        del tr
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "bench_step_parity.txt"), "a") as f:
        f.write("\n".join(report) + "\n")


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_step_with_the_fused_loss_tail_equals_the_step_on_the_materialised_noise(mode):
    """Round 6, the whole forward + backward (training_loop.py:376-378) with the loss tail inside the output conv and the noise kept as
    half-precision rows by the input conversion (Trainer's default with regenerated noise where ops.conv_loss_supported: B >= 16 at
    128x128) against the same step on the SAME noise handed over as a tensor (eps = half(c2w_philox_normal(seed)): separate input
    conversion, output conv, loss tail).  Same kernels in between, so the loss agrees to summation order and every conv weight gradient
    bit for bit; and the chain form of the residual blocks (outputs not written) against the written form within the mode's tolerance."""
    from climate2weather_amd.score import ScoreUNet
    from climate2weather_amd.training import Trainer
    B, C, H = 16, 65, 128
    torch.manual_seed(0)
    net = ScoreUNet(channels=C, spatial=2, activation=torch.nn.SiLU, **DEFAULT).cuda()
    gen = torch.Generator().manual_seed(16)
    x = (torch.randn(B, C, H, H, generator=gen) * 0.5 + 0.5).cuda()
    t = torch.rand(B, generator=gen).cuda()
    tr = Trainer(net, precision=mode, ema_rates=(), seed=3)
    tr.eng.chain_blocks = True  # opt-in: the first two runs use the chain form, the third the written form
    launches = []
    real = ops.conv

    def spy(*a, **k):
        launches.append((k.get("loss") is not None, bool(k.get("no_y")), k.get("resn") is not None))
        return real(*a, **k)
    ops.conv = spy
    try:
        tr.eng.flat_grad.zero_()
        l_fused = tr._forward_backward(x, t, None, sync=False).item()
        torch.cuda.synchronize()
        n_fused = list(launches)
        g_fused = tr.eng.flat_grad.clone()
        eps = torch.empty((B, C, H, H), device="cuda")
        ops.philox_normal(eps, eps.numel(), tr.last_noise_seed)
        eps = eps.half().float()
        launches.clear()
        tr.eng.flat_grad.zero_()
        l_plain = tr._forward_backward(x, t, eps, sync=False).item()
        torch.cuda.synchronize()
        n_plain = list(launches)
        g_plain = tr.eng.flat_grad.clone()
        tr.eng.chain_blocks = False
        launches.clear()
        tr.eng.flat_grad.zero_()
        l_written = tr._forward_backward(x, t, eps, sync=False).item()
        torch.cuda.synchronize()
        n_written = list(launches)
        g_written = tr.eng.flat_grad.clone()
    finally:
        ops.conv = real
    assert sum(1 for f in n_fused if f[0]) == 1 and not any(f[0] for f in n_plain)
    # outputs not written at B = 16: the 128^2 level only (1024 workgroups: the 16x16-tile kernel), descent 1 + ascent 1 (engine.run_blocks;
    # at the bench's B = 128 the 64^2 level adds descent 1 + ascent 2); each followed by a rebuilt residual
    assert sum(1 for f in n_plain if f[1]) == 2 and sum(1 for f in n_plain if f[2]) == 2 and not any(f[1] or f[2] for f in n_written)
    assert l_fused == pytest.approx(l_plain, rel=2e-5)
    S = tr.loss_scale()
    for n_, p in net.named_parameters():
        if p.dim() != 4:
            continue
        off, shape, _ = tr.eng.layout.views[n_]
        k = p.numel()
        assert torch.equal(g_fused[off:off + k], g_plain[off:off + k]), n_
    tol = 4e-2 if mode == "bf16" else 8e-3
    assert abs(l_plain - l_written) <= (3e-4 if mode == "bf16" else 3e-5) * l_written
    worst = (0.0, "")
    for n_, (off, shape, _) in tr.eng.layout.views.items():
        k = int(torch.tensor(shape).prod())
        a, b_ = g_plain[off:off + k] / S, g_written[off:off + k] / S
        worst = max(worst, ((a - b_).norm().item() / max(b_.norm().item(), 1e-30), n_))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "chain_vs_written_parity.txt"), "a") as f:
        f.write(f"B={B} C={C} {mode}: chain form vs written form, worst relative L2 over 228 gradient tensors {worst[0]:.2e} ({worst[1]}); "
                f"loss {l_plain:.6f} vs {l_written:.6f}; fused-loss step loss {l_fused:.6f}\n")
    assert worst[0] <= tol, worst


def test_two_stream_backward_reproduces_every_conv_weight_gradient_bit_for_bit():
    """Run-to-run determinism of the step as bench.py runs it (forward / input gradients on one stream, weight gradients and their
    reductions on a second): the same batch, (t, eps) and weights twelve times at B = 128.  The 70 conv kernels' gradients -- split-K
    partial sums reduced in a fixed order from deterministic operands -- must be bit-identical every time; a race between workgroups
    shows up here as one that changes (round 3's weight-ring race did, in the forward).  Biases, LayerNorm-modulation sums and the
    Linear weights fed by them use fp32 atomics and may differ in the last bits: not compared.  (The long-form hunts of round 3 are in the history: profiles/r03_experiments.md, "weight ring race".)"""
    from climate2weather_amd.score import ScoreUNet
    from climate2weather_amd.training import Trainer
    B, C, H = 128, 65, 128
    torch.manual_seed(0)
    net = ScoreUNet(channels=C, spatial=2, activation=torch.nn.SiLU, **DEFAULT).cuda()
    gen = torch.Generator().manual_seed(128)
    x = (torch.randn(B, C, H, H, generator=gen) * 0.5 + 0.5).cuda()
    t = torch.rand(B, generator=gen).cuda()
    eps = torch.randn(B, C, H, H, generator=gen).cuda()
    tr = Trainer(net, precision="bf16", ema_rates=())
    tr.eng.use_grad_stream = True  # both forms: the second queue (the default of rounds 1-4) is where a race between the streams would show
    convs = [(n, p) for n, p in net.named_parameters() if p.dim() == 4]
    assert len(convs) == 70
    ref = None
    for r in range(12):
        tr.eng.flat_grad.zero_()
        loss = tr._forward_backward(x, t, eps, sync=False)
        torch.cuda.synchronize()
        assert math.isfinite(loss.item())
        cur = [p.grad.detach().clone() for _, p in convs]
        if ref is None:
            ref = cur
            continue
        changed = [n for (n, _), a, b in zip(convs, cur, ref) if not torch.equal(a, b)]
        assert not changed, f"round {r}: {len(changed)} conv weight gradients changed between identical steps, first {changed[0]}"


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("windows", [False, True])
@pytest.mark.parametrize("C", [65, 52])
def test_input_conversion_that_keeps_its_noise_rows(dt, windows, C):
    """Round 6, c2w_nchw_to_nhwc_noise_rows (src/thor/pipelines.py:22-25): x_t = mu x + sigma eps with eps = the step's Philox stream
    rounded to half precision, and that eps kept as NHWC half rows for the loss tail.  Against the materialised stream
    (c2w_philox_normal) and tensor arithmetic; dense batches and windows read in place from a dataset array (img_off)."""
    B, H = 4, 128
    HW, ldc, lde = H * H, 128, (C + 7) // 8 * 8
    gen = torch.Generator(device=dev()).manual_seed(11)
    seed = 0x0123456789ABCDE
    musig = torch.rand((B, 2), generator=gen, device=dev()) + 0.25
    if windows:  # overlapping windows of 13 frames x F variables inside a (N, F, H, W) array, like data.DeviceWindowFeed hands them over
        F_ = C // 13
        data = torch.randn((40, F_, H, H), generator=gen, device=dev())
        starts = torch.tensor([3, 17, 0, 26], device=dev())
        offs = (starts * F_ * HW).to(torch.int64)
        x = torch.stack([data[s:s + 13].reshape(C, H, H) for s in starts.tolist()])
        src = data
    else:
        x = torch.randn((B, C, H, H), generator=gen, device=dev())
        src, offs = x, None
    y = torch.full((B * HW, ldc), 5.0, dtype=TD[dt], device=dev())
    er = torch.full((B * HW, lde), 5.0, dtype=torch.float16, device=dev())
    assert ops.nchw_to_nhwc_noise_rows(src, offs, seed, musig, y, er, B, C, HW, ldc, lde, dt)
    eps = torch.empty((B, C, H, H), device=dev())
    ops.philox_normal(eps, eps.numel(), seed)
    e16 = eps.to(torch.float16)
    assert torch.equal(er[:, :C], e16.permute(0, 2, 3, 1).reshape(B * HW, C))
    assert er[:, C:].float().abs().max().item() == 0.0 if lde > C else True
    xt = torch.addcmul(musig[:, 0].view(B, 1, 1, 1) * x, musig[:, 1].view(B, 1, 1, 1), e16.float())  # fma(sigma, eps, mu * x)
    exp = xt.permute(0, 2, 3, 1).reshape(B * HW, C).to(TD[dt])
    same = (y[:, :C] == exp).float().mean().item()
    assert same >= 0.999, same  # (torch's addcmul may or may not fuse: a last-bit difference before the 16-bit rounding flips few values)
    close(y[:, :C], exp, 2.0 ** -7 if dt == BF16 else 2.0 ** -10, "x_t rows")
    assert y[:, C:].float().abs().max().item() == 0.0


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("C,scaled", [(65, False), (52, True), (80, False)])
def test_loss_tail_fused_into_the_output_convolution(dt, packed, C, scaled):
    """Round 6: c2w_conv_forward with C2wConvArgs.loss_* (src/thor/pipelines.py:35, training_loop.py:376-377): the network-output conv
    (model/nn.py:194; narrow form of the 16x16-tile kernel, >= 1024 workgroups) turns its tile into dY = (prediction - eps) * gscale and
    adds sum (prediction - eps)^2 to loss_sum, eps read from the half-precision noise rows.  Against the unfused pair of launches it
    replaces -- the same conv followed by c2w_mse_loss_grad on the same noise -- and the same arithmetic in PyTorch on the stored prediction."""
    B, H = 16, 128
    g = geom(B, H, H, 128, H, H, 128, 128, C, S1)
    assert ops.conv_loss_supported(g, dt) and ops.conv_dispatch(g, dt) == T3
    npix, lde = B * H * H, (C + 7) // 8 * 8
    x = rnd((npix, 128), dt, 1)
    w = rnd((C, 9, 128), dt, 2, scale=1.0 / math.sqrt(9 * 128))
    bias = rnd((C,), F32, 3)
    wop = w
    if packed:
        wop = torch.empty(ops.packed_conv_weights_numel(C, 128), dtype=TD[dt], device=dev())
        ops.pack_conv_weights_batched(w, wop, torch.tensor([[0, 0, C, 128]], dtype=torch.int64, device=dev()), 1, dt)
    er = torch.zeros((npix, lde), dtype=torch.float16, device=dev())
    er[:, :C] = rnd((npix, C), F16, 9)
    gs = 2.0 / (B * C * H * H)
    scaler = torch.tensor([512.0, 0.0, 0.0, 0.0], device=dev()) if scaled else None
    # unfused: conv, then the loss tail on the same noise (as an NCHW fp32 tensor)
    y = torch.empty((npix, 128), dtype=TD[dt], device=dev())
    ops.conv(x, wop, bias, y, g, dt, wpacked=packed)
    eps_nchw = er[:, :C].float().view(B, H * H, C).permute(0, 2, 1).contiguous()
    dy_ref, ls_ref = torch.full_like(y, 9.0), torch.zeros(1, device=dev())
    ops.mse_loss_grad(y, eps_nchw, dy_ref, ls_ref, B, C, H * H, 128, gs, dt, scaler=scaler)
    # fused
    dy, ls = torch.full_like(y, 7.0), torch.zeros(1, device=dev())
    lf = dict(sum=ls, eps=er, lde=lde, gscale=gs, C=C, scaler=scaler)
    ops.conv(x, wop, bias, dy, g, dt, wpacked=packed, loss=lf)
    torch.cuda.synchronize()
    assert torch.equal(dy, dy_ref)
    assert dy[:, :C].float().abs().sum().item() > 0 and dy[:, C:].float().abs().max().item() == 0.0  # padding channels: zero gradient
    assert ls.item() == pytest.approx(ls_ref.item(), rel=2e-5)  # 17 M squares summed in another order (fp32)
    d = y[:, :C].float() - er[:, :C].float()
    assert ls.item() == pytest.approx((d.double() ** 2).sum().item(), rel=2e-5)
    assert torch.equal(dy[:, :C], (d * (gs * (512.0 if scaled else 1.0))).to(TD[dt]))
    # a second launch ACCUMULATES into loss_sum (the trainer zeroes it per round)
    ops.conv(x, wop, bias, dy, g, dt, wpacked=packed, loss=lf)
    assert ls.item() == pytest.approx(2 * ls_ref.item(), rel=2e-5)


def test_loss_fusion_is_refused_where_the_kernel_does_not_exist():
    """No silent unfused result: geometries outside c2w_conv_loss_supported fail loudly when handed loss arguments."""
    g_small = geom(2, 128, 128, 128, 128, 128, 128, 128, 65, S1)  # 128 workgroups: the 8x16-tile kernel
    g_wide = geom(16, 128, 128, 128, 128, 128, 128, 128, 128, S1)  # a full-width conv
    for g in (g_small, g_wide):
        assert not ops.conv_loss_supported(g, BF16)
        npix = g["B"] * 128 * 128
        x, w = rnd((npix, 128), BF16, 1), rnd((g["wrows"], 9, 128), BF16, 2, scale=0.03)
        y, ls = torch.empty((npix, 128), dtype=torch.bfloat16, device=dev()), torch.zeros(1, device=dev())
        er = torch.zeros((npix, 72), dtype=torch.float16, device=dev())
        with pytest.raises(_lib.C2wError):
            ops.conv(x, w, None, y, g, BF16, loss=dict(sum=ls, eps=er, lde=72, gscale=1.0, C=min(65, g["wrows"])))
    assert not ops.conv_loss_supported(geom(16, 128, 128, 128, 128, 128, 128, 128, 65, S1), F32)
    g_ok = geom(16, 128, 128, 128, 128, 128, 128, 128, 65, S1)
    npix = 16 * 128 * 128
    x, w = rnd((npix, 128), BF16, 1), rnd((65, 9, 128), BF16, 2, scale=0.03)
    y, ls = torch.empty((npix, 128), dtype=torch.bfloat16, device=dev()), torch.zeros(1, device=dev())
    with pytest.raises(_lib.C2wError):  # a channel stride that is not whole 16-byte segments
        ops.conv(x, w, None, y, g_ok, BF16, loss=dict(sum=ls, eps=torch.zeros((npix, 68), dtype=torch.float16, device=dev()), lde=68, gscale=1.0, C=65))


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("form", ["plain residual, output not written", "rebuilt residual, output not written", "rebuilt residual, output written",
                                  "rebuilt residual, plain LayerNorm (up-block consumer)"])
def test_chain_form_of_the_layernorm_emitting_convolution(form, packed, dt):
    """Round 6 (C2wConvArgs.lnf_mean / res_rstd + res_mean + res_m / C2W_CONV_NO_Y; model/nn.py:27-28,146-159): a residual block's second
    conv on the 16x16-tile kernel emits the next block's LayerNorm input with its mean and 1/sigma, may leave its own output
    unwritten, and may rebuild its residual x = h / rstd + mean - m from what an earlier launch emitted.  Against the PyTorch
    restatement of exactly that arithmetic (tests/emu_ops.py::conv, chain branch) on the same operands; the rebuilt residual itself
    against the tensor it replaces."""
    B, H, C = 16, 128, 128
    g = geom(B, H, H, C, H, H, C, C, C, S1)
    assert ops.conv_lnfwd_chain_supported(g, dt) and ops.conv_dispatch(g, dt, fused_ln=True) == T3
    npix = B * H * H
    x = rnd((npix, C), dt, 1)
    w = rnd((C, 9, C), dt, 2, scale=1.0 / math.sqrt(9 * C))
    bias = rnd((C,), F32, 3)
    wop = w
    if packed:
        wop = torch.empty(ops.packed_conv_weights_numel(C, C), dtype=TD[dt], device=dev())
        ops.pack_conv_weights_batched(w, wop, torch.tensor([[0, 0, C, C]], dtype=torch.int64, device=dev()), 1, dt)
    ldm = C + 64
    m_next = rnd((B, ldm), F32, 4).view(-1)
    rebuilt, no_y, plain_ln = form.startswith("rebuilt"), "not written" in form, "plain LayerNorm" in form
    # the residual: a block input x_k, and what an earlier launch would have emitted for it: h = LN(x_k + m_k), mean, 1/sigma
    xk = rnd((npix, C), dt, 5, scale=2.0)
    m_k = rnd((B, ldm), F32, 6).view(-1)
    xm = xk.float() + E._mrows(m_k, npix, H * H, C, ldm)
    var, mean = torch.var_mean(xm, dim=1, unbiased=True, keepdim=True)
    rs = (var + 1e-5).rsqrt()
    hk = ((xm - mean) * rs).to(TD[dt])
    resn = dict(rstd=rs.view(-1).contiguous(), mean=mean.view(-1).contiguous(), m=m_k) if rebuilt else None
    outs = {}
    for which, conv in (("hip", ops.conv), ("ref", E.conv)):
        y = torch.full((npix, C), 5.0, dtype=TD[dt], device=dev())
        hn = torch.full((npix, C), 3.0, dtype=TD[dt], device=dev())
        lnf = dict(y=hn, m=None if plain_ln else m_next, ldm=0 if plain_ln else ldm, eps=1e-5, unbiased=True)
        if not plain_ln:
            lnf["rstd"] = torch.zeros(npix, device=dev())
        if no_y:
            lnf["mean"] = torch.zeros(npix, device=dev())
        kw = dict(wpacked=packed) if which == "hip" else {}
        if which == "ref":
            E.CHAIN = True
        if resn is not None:  # (the modulation rows of both LayerNorms share one stride, like the engine's m_all)
            conv(x, wop if which == "hip" else w, bias, y, g, dt, res=hk, lnf=lnf, resn=resn, no_y=no_y, **kw)
        else:
            conv(x, wop if which == "hip" else w, bias, y, g, dt, res=xk, lnf=lnf, no_y=no_y, **kw)
        torch.cuda.synchronize()
        outs[which] = (y, hn, lnf.get("rstd"), lnf.get("mean"))
    (y, hn, rstd, mu), (y_r, hn_r, rstd_r, mu_r) = outs["hip"], outs["ref"]
    close(hn, hn_r, TOL[dt], form + ": emitted LayerNorm")
    if rstd is not None:
        close(rstd, rstd_r, 2e-3, form + ": 1/sigma")
    if no_y:
        assert torch.all(y == 5.0)  # not a byte of the output buffer was touched
        close(mu, mu_r, 2e-3, form + ": mean")
    else:
        close(y, y_r, TOL[dt], form + ": output")
    if rebuilt:  # the rebuilt residual against the tensor it stands for: h carries the storage type's rounding relative to sigma
        xr = hk.float() / rs + mean - E._mrows(m_k, npix, H * H, C, ldm)
        close(xr, xk, 1.5 * TOL[dt] / 4, form + ": rebuilt residual vs the block input it replaces")


def test_chain_fields_are_refused_where_the_kernel_does_not_exist():
    """lnf_mean / res_rstd / C2W_CONV_NO_Y outside c2w_conv_lnfwd_chain_supported: a loud failure, never a silently written output."""
    g = geom(2, 128, 128, 128, 128, 128, 128, 128, 128, S1)  # 128 workgroups: the 8x16-tile kernel (its LayerNorm emission has no chain form)
    assert ops.conv_lnfwd_supported(g, BF16) and not ops.conv_lnfwd_chain_supported(g, BF16)
    npix = 2 * 128 * 128
    x, w = rnd((npix, 128), BF16, 1), rnd((128, 9, 128), BF16, 2, scale=0.03)
    y, hn = torch.empty((npix, 128), dtype=torch.bfloat16, device=dev()), torch.empty((npix, 128), dtype=torch.bfloat16, device=dev())
    lnf = dict(y=hn, m=None, ldm=0, eps=1e-5, unbiased=True, rstd=torch.zeros(npix, device=dev()), mean=torch.zeros(npix, device=dev()))
    with pytest.raises(_lib.C2wError):
        ops.conv(x, w, None, y, g, BF16, res=x, lnf=lnf, no_y=True)
    with pytest.raises(_lib.C2wError):  # statistics without the rows they belong to
        ops.conv(x, w, None, y, geom(16, 128, 128, 128, 128, 128, 128, 128, 128, S1), BF16, lnf=dict(lnf, mean=None),
                 resn=dict(rstd=torch.zeros(npix, device=dev()), mean=torch.zeros(npix, device=dev())))


PACKED_CASES = [  # (name, B, H, Cin, Cout, wrows, epilogue)
    ("128->128 @128^2 bias+SiLU pair", 16, 128, 128, 128, 128, "pair"),
    ("128->128 @128^2 residual + LayerNorm emission", 16, 128, 128, 128, 128, "res+lnf"),
    ("256->256 @32^2 (two 128-row tiles) multiplier + residual", 128, 32, 256, 256, 256, "mulp+res"),
    ("256->128 @64^2", 64, 64, 256, 128, 128, "bias"),
    ("output conv 128->65(128) @128^2 (narrow form)", 16, 128, 128, 128, 65, "bias"),
    ("padded K: 160(192)->128 @128^2 with kvalid", 16, 128, 192, 128, 128, "kvalid"),
]


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("case", PACKED_CASES, ids=[c[0] for c in PACKED_CASES])
def test_stage_major_packed_weights_give_the_same_bits(case, dt):
    """C2W_CONV_WPACKED: the 16x16-tile kernel fed the stage-major copy of the weights (c2w_pack_conv_weights_batched) must reproduce
    the launch with the plain [rows][9][Cin] weights bit for bit -- same arithmetic, another address for the same bytes -- over the
    epilogue families, two-tile output rows, the narrow output-conv form and a kvalid launch; and the launches that cannot take it
    (fused LayerNorm backward) must say so."""
    name, B, H, Cin, Cout, wrows, ep = case
    g = geom(B, H, H, Cin, H, H, Cout, Cout, wrows, S1)
    assert ops.conv_wpacked_supported(g, dt), name
    npix = B * H * H
    x = rnd((npix, Cin), dt, 1)
    w = rnd((wrows, 9, Cin), dt, 2, scale=1.0 / math.sqrt(9 * Cin))
    kvalid = 0
    if ep == "kvalid":
        kvalid = 160
        x[:, kvalid:] = 0
    wp = torch.empty(ops.packed_conv_weights_numel(wrows, Cin) + 64, dtype=TD[dt], device=dev()).fill_(7.0)
    desc = torch.tensor([[0, 0, wrows, Cin]], dtype=torch.int64, device=dev())
    ops.pack_conv_weights_batched(w, wp, desc, 1, dt)
    assert torch.all(wp[-64:] == 7.0)  # nothing written past the packed matrix
    bias = rnd((wrows,), F32, 3)
    kw = {}
    outs = []
    for packed in (False, True):
        y = torch.full((npix, Cout), 5.0, dtype=TD[dt], device=dev())
        kw = dict(kvalid=kvalid)
        extra = []
        if ep == "pair":
            y2 = torch.full_like(y, 3.0)
            kw.update(act=ops.ACT_SILU_PAIR, y2=y2)
            extra.append(y2)
        if "res" in ep:
            kw["res"] = rnd((npix, Cout), dt, 4)
        if "mulp" in ep:
            kw.update(mul=rnd((npix, Cout), dt, 5), mulmode=ops.MUL_PLAIN)
        if "lnf" in ep:
            hn = torch.full_like(y, 3.0)
            kw["lnf"] = dict(y=hn, m=rnd((B, Cout), F32, 6).view(-1), ldm=Cout, eps=1e-5, unbiased=True)
            extra.append(hn)
        ops.conv(x, wp if packed else w, bias, y, g, dt, wpacked=packed, **kw)
        torch.cuda.synchronize()
        outs.append([y] + extra)
    for a, b in zip(*outs):
        assert a.float().abs().sum().item() > 0 and torch.equal(a, b), name
    # the fused LayerNorm backward (its packed instantiation recomputes a fragment base per stage to stay inside 128 registers)
    if Cout == 128 and wrows == 128 and Cin == 128:
        xs, ms, rs = rnd((npix, Cout), dt, 7), rnd((B, Cout), F32, 8), rnd((npix, Cout), dt, 4)
        got = []
        for packed in (False, True):
            dm = torch.zeros(B, Cout, device=dev())
            y = torch.full((npix, Cout), 5.0, dtype=TD[dt], device=dev())
            ops.conv(x, wp if packed else w, None, y, g, dt, res=rs, ln=dict(x=xs, m=ms.view(-1), dm=dm.view(-1), ldm=Cout, eps=1e-5, unbiased=True),
                     wpacked=packed)
            torch.cuda.synchronize()
            got.append(y)
        assert torch.equal(got[0], got[1]), name + " (fused LayerNorm backward)"


SPLITK_CASES = [  # (name, pair, B, H, Cin, Cout, expected workgroups per tile)
    ("512->512 @8x8, 37 windows (L = 49): 76 tiles, chunks 2 / 3 / 3", True, 37, 8, 512, 512, 3),
    ("384->384 @16x16, 18 windows: 108 tiles", False, 18, 16, 384, 384, 2),
    ("512->512 @8x8, 5 windows: 12 tiles, every chunk its own workgroup", True, 5, 8, 512, 512, 8),
    ("256->256 @32x32, 3 windows: 48 tiles", False, 3, 32, 256, 256, 4),
]


@pytest.mark.parametrize("dt", [BF16, F16, F32])
@pytest.mark.parametrize("epi", ["bias+silu", "bias+res", "mul(dsilu)+res", "plain"])
@pytest.mark.parametrize("case", SPLITK_CASES, ids=[c[0] for c in SPLITK_CASES])
def test_split_k_convolution_for_underfilled_launches(case, epi, dt):
    """Round 6 (C2wConvArgs.splitk; model/nn.py:146-159 at the deep levels of a sampler step on a short trajectory, exp/configs/000_on-model-eval/
    s16_t6.yml): a conv launch with fewer output tiles than CUs deals its K chunks to c2w_conv_splitk_plan workgroups per tile and a second
    launch adds the partial tiles in a fixed order.  Against the unsplit launch of the same arguments (summation order differs: the
    storage type's rounding) and the PyTorch restatement; twice for bit-reproducibility; the plan's answers; refusal of a wrong plan."""
    name, pair, B, H, Cin, Cout, want = case
    if dt == F32:
        Cin = Cin // 2  # (fp32 chunks are 32 channels: the same chunk counts)
    g = geom(B, H, H, Cin, H, H, Cout, Cout, Cout, S1)
    act = ops.ACT_SILU if "silu" in epi and "dsilu" not in epi else ops.ACT_NONE
    ns, nbytes = ops.conv_splitk_plan(g, dt, act)
    assert ns == want and nbytes == ns * ((B + 1) // 2 * (H // 8) if pair else B * (H // 8) * (H // 16)) * ((Cout + 127) // 128) * 65536, (ns, nbytes)
    assert ops.conv_dispatch(g, dt) == (PAIR if pair else HALF)
    npix = B * H * H
    x = rnd((npix, Cin), dt, 1)
    w = rnd((Cout, 9, Cin), dt, 2, scale=1.0 / math.sqrt(9 * Cin))
    bias = rnd((Cout,), F32, 3) if "bias" in epi else None
    kw = dict(act=act)
    if "res" in epi:
        kw["res"] = rnd((npix, Cout), dt, 4)
    if "mul" in epi:
        kw.update(mul=rnd((npix, Cout), dt, 5), mulmode=ops.MUL_DSILU)
    ws = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=dev()).fill_(float("nan"))
    y0 = torch.full((npix, Cout), 5.0, dtype=TD[dt], device=dev())
    ops.conv(x, w, bias, y0, g, dt, **kw)
    outs = []
    for _ in range(2):
        y = torch.full((npix, Cout), 7.0, dtype=TD[dt], device=dev())
        ws.fill_(float("nan"))
        ops.conv(x, w, bias, y, g, dt, splitk=(ws, ns), **kw)
        torch.cuda.synchronize()
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    assert torch.isnan(ws[nbytes // 4:]).all()  # nothing written past the plan's bytes
    tol = 1e-5 if dt == F32 else TOL[dt] / 4
    close(outs[0], y0, tol, name + " / " + epi + ": split vs unsplit")
    y_ref = torch.empty_like(y0)
    E.conv(x, w, bias, y_ref, g, dt, **kw)
    close(outs[0], y_ref, 2e-5 if dt == F32 else TOL[dt], name + " / " + epi + ": split vs restatement")
    with pytest.raises(_lib.C2wError):  # not the plan's answer
        ops.conv(x, w, bias, y, g, dt, splitk=(ws, ns + 1 if ns < 8 else 2), **kw)  # (scratch sized for the plan's answer; the count alone is refused)
    with pytest.raises(_lib.C2wError):  # scratch too small
        ops.conv(x, w, bias, y, g, dt, splitk=(ws[: nbytes // 8], ns), **kw)


def test_split_k_plan_leaves_filled_launches_alone():
    for g, dt in ((geom(128, 8, 8, 512, 8, 8, 512, 512, 512, S1), BF16),      # 256 tiles: a workgroup per CU
                  (geom(37, 16, 16, 384, 16, 16, 384, 384, 384, S1), BF16),   # 222 tiles: two per tile would be two workgroups on most CUs
                  (geom(37, 64, 64, 128, 64, 64, 128, 128, 128, S1), BF16),   # 592 tiles on the 16x16-tile kernel
                  (geom(4, 16, 16, 64, 16, 16, 128, 128, 128, S1), BF16)):    # one K chunk: nothing to split
        assert ops.conv_splitk_plan(g, dt) == (1, 0)


@pytest.mark.parametrize("dt", [BF16, F16, F32])
@pytest.mark.parametrize("case", [("512->512 @8x8 paired, B = 128: 256 workgroups", 128, 8, 512, 512),
                                  ("384->384 @16x16, B = 32: 192 workgroups", 32, 16, 384, 384),
                                  ("256->256 @32x32, B = 4: 64 workgroups, residual + multiplier", 4, 32, 256, 256),
                                  ("384->384 @16x16, B = 96: 576 workgroups (one patch buffer, two workgroups per CU)", 96, 16, 384, 384),
                                  ("1024->512 @8x8 paired, B = 24: 48 workgroups, sixteen K chunks", 24, 8, 1024, 512)],
                         ids=["pair", "16x16", "32x32", "16x16-576wg", "pair-16chunks"])
def test_eight_wave_tile_kernel_gives_the_four_wave_kernels_bits(case, dt, monkeypatch):
    """Round 6, conv_patch_half8_kernel (launches of at most one workgroup per CU): the same tile, the same K order per output element
    (chunks, taps, the two K halves of a stage) on eight waves instead of four -- every output bit for bit what C2W_NO_HALF8=1 gives,
    over the epilogue flavours the 8x16-tile kernels carry (bias + SiLU pair, multiplier + residual, plain) and the split-K form; and
    the same bits with one patch buffer (C2W_HALF8_DB=0) as with the two that launches of at most 256 workgroups get by default."""
    name, B, H, Cin, Cout = case
    if dt == F32:
        Cin = Cin // 2
    g = geom(B, H, H, Cin, H, H, Cout, Cout, Cout, S1)
    assert ops.conv_dispatch(g, dt) in (PAIR, HALF)
    npix = B * H * H
    x = rnd((npix, Cin), dt, 1)
    w = rnd((Cout, 9, Cin), dt, 2, scale=1.0 / math.sqrt(9 * Cin))
    bias = rnd((Cout,), F32, 3)
    res, mul = rnd((npix, Cout), dt, 4), rnd((npix, Cout), dt, 5)
    outs = {}
    for half8 in (True, "one patch buffer", False):
        if half8 == "one patch buffer":
            monkeypatch.setenv("C2W_HALF8_DB", "0")
        if not half8:
            monkeypatch.delenv("C2W_HALF8_DB")
            monkeypatch.setenv("C2W_NO_HALF8", "1")
        ops.knobs_reload()
        got = []
        y, y2 = torch.full((npix, Cout), 5.0, dtype=TD[dt], device=dev()), torch.full((npix, Cout), 3.0, dtype=TD[dt], device=dev())
        ops.conv(x, w, bias, y, g, dt, act=ops.ACT_SILU_PAIR, y2=y2)
        got += [y, y2]
        y = torch.full((npix, Cout), 5.0, dtype=TD[dt], device=dev())
        ops.conv(x, w, None, y, g, dt, mul=mul, mulmode=ops.MUL_DSILU, res=res)
        got.append(y)
        y = torch.full((npix, Cout), 5.0, dtype=TD[dt], device=dev())
        ops.conv(x, w, bias, y, g, dt)
        got.append(y)
        ns, nbytes = ops.conv_splitk_plan(g, dt)
        if ns > 1:
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev())
            y = torch.full((npix, Cout), 5.0, dtype=TD[dt], device=dev())
            ops.conv(x, w, bias, y, g, dt, res=res, splitk=(ws, ns))
            got.append(y)
        torch.cuda.synchronize()
        outs[half8] = got
    monkeypatch.delenv("C2W_NO_HALF8")
    ops.knobs_reload()
    assert len(outs[True]) == len(outs[False]) == len(outs["one patch buffer"])
    for i, (a, a1, b_) in enumerate(zip(outs[True], outs["one patch buffer"], outs[False])):
        assert a.float().abs().sum().item() > 0 and torch.equal(a, b_) and torch.equal(a1, b_), (name, i)


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("case", [("128->128 from 16x16 (two chunks, one channel tile), residual", 8, 16, 128, 128),
                                  ("256->128 from 8x16 (four chunks), residual", 16, 8, 256, 128),
                                  ("128->256 from 32x32 (two channel tiles), no residual", 4, 32, 128, 256)], ids=["a", "b", "c"])
def test_two_class_stride2_input_gradient_gives_the_one_class_kernels_bits(case, dt, monkeypatch):
    """Round 6, conv_patch_ts2_pairs_kernel (model/nn.py:169-174 backward): two output-parity classes per workgroup, each class's taps in the
    order the one-class kernel walks them -- every output bit for bit what C2W_TS2_PAIRS=0 gives (the fp16 form runs its stages without
    the deferred K half: the same order of accumulation)."""
    name, B, Hd, Cdy, Cdx = case
    Wd = 16 if Hd == 8 else Hd
    g = geom(B, Hd, Wd, Cdy, 2 * Hd, 2 * Wd, Cdx, Cdx, Cdx, TS2)
    assert ops.conv_dispatch(g, dt) == TS2P
    dy = rnd((B * Hd * Wd, Cdy), dt, 1)
    wt = rnd((Cdx, 9, Cdy), dt, 2, scale=1.0 / math.sqrt(9 * Cdy))
    res = rnd((B * 4 * Hd * Wd, Cdx), dt, 3) if "no residual" not in name else None
    outs = []
    for pairs in (True, False):
        if not pairs:
            monkeypatch.setenv("C2W_TS2_PAIRS", "0")
        ops.knobs_reload()
        dx = torch.full((B * 4 * Hd * Wd, Cdx), 5.0, dtype=TD[dt], device=dev())
        ops.conv(dy, wt, None, dx, g, dt, res=res)
        torch.cuda.synchronize()
        outs.append(dx)
    monkeypatch.delenv("C2W_TS2_PAIRS")
    ops.knobs_reload()
    assert outs[0].float().abs().sum().item() > 0 and torch.equal(outs[0], outs[1]), name
    ref = torch.empty_like(outs[0])
    E.conv(dy, wt, None, ref, g, dt, res=res)
    close(outs[0], ref, TOL[dt], name)
