"""TEST DOUBLE for climate2weather_amd.ops: every launcher re-stated with plain PyTorch ops on the same buffers.

Two uses, both inside tests/ only:
  * CPU (`-m "not gpu"`): monkeypatched over `climate2weather_amd.ops` so the engine's orchestration (buffer wiring,
    flat parameter layout, the hand-written backward tape, the sampler) is checked against the oracle without a GPU;
  * GPU (`-m gpu`): the per-kernel "plain PyTorch fp32 reference of the same op" each HIP kernel is compared with.
The product never imports this module.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
CONV_1X1, CONV_S1, CONV_S2, CONV_UP, CONV_TS2 = 0, 1, 2, 3, 4
ACT_NONE, ACT_SILU, ACT_SILU_PAIR, ACT_RELU, ACT_RELU_PAIR = 0, 1, 2, 3, 4
MUL_PLAIN, MUL_DSILU = 0, 1
TD = {DTYPE_F32: torch.float32, DTYPE_BF16: torch.bfloat16, DTYPE_F16: torch.float16}


def _rows(t, n, ld):
    return t.reshape(-1)[: n * ld].view(n, ld)


def _dsilu(a):
    s = torch.sigmoid(a)
    return s * (1 + a * (1 - s))


def _conv_core(X, Wt, g):
    """X: (B,Cin,Hin,Win) fp32; Wt: (rows,taps,Cin) fp32 -> (B,rows,Hout,Wout)"""
    mode = g["mode"]
    rows = Wt.shape[0]
    if mode == CONV_1X1:
        return torch.einsum("bchw,oc->bohw", X, Wt[:, 0])
    W4 = Wt.view(rows, 3, 3, -1).permute(0, 3, 1, 2)
    if mode == CONV_S1:
        return F.conv2d(X, W4, padding=1)
    if mode == CONV_S2:
        return F.conv2d(X, W4, stride=2, padding=1)
    if mode == CONV_UP:
        return F.conv2d(F.interpolate(X, scale_factor=2.0, mode="nearest"), W4, padding=1)
    if mode == CONV_TS2:
        op = (g["Hout"] - (2 * g["Hin"] - 1), g["Wout"] - (2 * g["Win"] - 1))
        return F.conv_transpose2d(X, W4.permute(1, 0, 2, 3), stride=2, padding=1, output_padding=op)
    raise ValueError(mode)


def conv_patch_supported(g, dtype):
    if g["mode"] == CONV_UP:
        return g["Hout"] == 2 * g["Hin"] and g["Hout"] % 8 == 0 and g["Wout"] % 16 == 0
    return g["mode"] == CONV_S1 and g["Hin"] == g["Hout"] and g["Hin"] % 8 == 0 and g["Win"] % 16 == 0


def upsample2(x, y, B, H, W, C, dtype):
    X = _rows(x, B * H * W, C).view(B, H, W, C)
    _rows(y, B * 4 * H * W, C)[:] = X.repeat_interleave(2, 1).repeat_interleave(2, 2).reshape(-1, C)


def conv_lnbwd_supported(g, dtype):
    return g["mode"] == CONV_S1 and g["Cout"] == g["ldy"]  # the emulation fuses wherever the semantics are defined


def conv_lnfwd_supported(g, dtype):
    return g["mode"] in (CONV_S1, CONV_UP) and g["Cout"] == g["ldy"]


CHAIN = True  # tests flip it: does the emulation offer the chain form (residual blocks whose intermediate outputs are never written)?


def conv_lnfwd_chain_supported(g, dtype):
    return CHAIN and conv_lnfwd_supported(g, dtype) and g["mode"] == CONV_S1 and dtype != DTYPE_F32


def conv_splitk_plan(g, dtype, act=ACT_NONE):
    return 1, 0  # summation order is the kernels' business


def conv_loss_supported(g, dtype):
    return False  # the emulation has no Philox stream: trainers on the CPU inject eps and run the separate loss tail


def conv_pool2_supported(g, dtype):
    return conv_patch_supported(g, dtype) and g["mode"] == CONV_S1 and g["Win"] != 8


PACKED_FROM_PIXELS = None  # tests set a pixel count: 3x3 launches with at least that many output pixels "take packed weights"


def conv_wpacked_supported(g, dtype):
    """Off by default: the emulation reads the plain [rows][taps][Cin] weights.  With PACKED_FROM_PIXELS set it exercises the engine's
    packed-copy bookkeeping (offsets, versions, stable addresses): pack_conv_weights_batched below stores every row XOR-ed with a tag,
    which the emulated conv undoes when it is told the operand is the packed copy -- a plain operand passed as packed (or a stale
    packed copy) gives wrong numbers."""
    return PACKED_FROM_PIXELS is not None and g["mode"] in (CONV_S1, CONV_UP) and dtype != DTYPE_F32 and \
        g["B"] * g["Hout"] * g["Wout"] >= PACKED_FROM_PIXELS


def packed_conv_weights_numel(rows, cin):
    return 9 * cin * ((rows + 127) // 128 * 128)


def pack_conv_weights_batched(src, dst, desc, n, dtype):
    for so, do, rows, k in desc.view(-1, 4)[:n].tolist():
        m = rows * 9 * k
        dst.reshape(-1)[do:do + m] = -src.reshape(-1)[so:so + m]  # "layout" of the emulated packed copy: the negated matrix


def conv(x, w, bias, y, g, dtype, act=ACT_NONE, res=None, mul=None, mulmode=MUL_PLAIN, naive=False, y2=None, ln=None, lnf=None, pool2=False, kvalid=0,
         wpacked=False, loss=None, resn=None, no_y=False, splitk=None):
    assert loss is None, "conv_loss_supported() is False here: nobody may ask the emulation for the fused loss"
    assert splitk is None, "conv_splitk_plan() answers (1, 0) here"
    if wpacked:
        assert conv_wpacked_supported(g, dtype)
        return conv(x, -w.reshape(-1)[: g["wrows"] * 9 * g["Cin"]], bias, y, g, dtype, act, res, mul, mulmode, naive, y2, ln, lnf, pool2, kvalid,
                    resn=resn, no_y=no_y)
    # kvalid: a promise that input channels >= kvalid are zero (the HIP kernels may skip them); the restatement multiplies everything
    if resn is not None or no_y or (lnf is not None and lnf.get("mean") is not None):
        # the chain form (c2w_hip.h, round 6): residual rebuilt from normalised rows, the sum not written, the LayerNorm's mean kept
        assert lnf is not None and ln is None and mul is None and y2 is None and act == ACT_NONE and conv_lnfwd_chain_supported(g, dtype)
        npix, HW, C, T = g["B"] * g["Hout"] * g["Wout"], g["Hout"] * g["Wout"], g["Cout"], TD[dtype]
        tmp = torch.zeros((npix, g["ldy"]), dtype=T, device=x.device)
        conv(x, w, bias, tmp, g, dtype)
        u = tmp[:, :C].float()  # the conv result rounded to the storage type (the tile in LDS)
        if resn is not None:
            hrows = _rows(res, npix, g["ldy"])[:, :C].float()
            sig = 1.0 / resn["rstd"].reshape(-1)[:npix].float().unsqueeze(1)
            u = u + (hrows * sig + (resn["mean"].reshape(-1)[:npix].float().unsqueeze(1) - _mrows(resn.get("m"), npix, HW, C, lnf.get("ldm", 0))))
        elif res is not None:
            u = u + _rows(res, npix, g["ldy"])[:, :C].float()
        if not no_y:
            _rows(y, npix, g["ldy"])[:, :C] = u.to(T)
            u = u.to(T).float()
        xm = u + _mrows(lnf.get("m"), npix, HW, C, lnf.get("ldm", 0))
        var, mean = torch.var_mean(xm, dim=1, unbiased=bool(lnf["unbiased"]), keepdim=True)
        rs = (var + lnf["eps"]).rsqrt()
        _rows(lnf["y"], npix, g["ldy"])[:, :C] = ((xm - mean) * rs).to(T)
        if lnf.get("rstd") is not None:
            lnf["rstd"].reshape(-1)[:npix] = rs.view(-1)
        if lnf.get("mean") is not None:
            lnf["mean"].reshape(-1)[:npix] = mean.view(-1)
        return
    if lnf is not None:  # second output: LN of the stored result (+ the consumer's modulation)
        assert ln is None and mul is None and y2 is None and act == ACT_NONE
        conv(x, w, bias, y, g, dtype, res=res)
        npix = g["B"] * g["Hout"] * g["Wout"]
        ln_forward(y, lnf.get("m"), lnf["y"], npix, g["Hout"] * g["Wout"], g["Cout"], lnf.get("ldm", 0), lnf["eps"], lnf["unbiased"], dtype)
        if lnf.get("rstd") is not None:  # every pixel row's 1/sigma, as the normalisation used it
            xm = _rows(y, npix, g["ldy"])[:, : g["Cout"]].float() + _mrows(lnf.get("m"), npix, g["Hout"] * g["Wout"], g["Cout"], lnf.get("ldm", 0))
            var = xm.var(dim=1, unbiased=bool(lnf["unbiased"]))
            lnf["rstd"].reshape(-1)[:npix] = (var + lnf["eps"]).rsqrt()
        return
    if ln is not None:  # y = res + dLN(conv(x); ln.x + ln.m), the conv result rounded to the storage type in between
        assert mul is None and y2 is None and act == ACT_NONE
        tmp = torch.zeros_like(y)
        conv(x, w, bias, tmp, g, dtype)
        npix = g["B"] * g["Hout"] * g["Wout"]
        if ln.get("rstd") is not None:  # ln["x"] = the normalised rows the forward kept, ln["rstd"] their 1/sigma
            C = g["Cout"]
            gq = _rows(tmp, npix, g["ldy"])[:, :C].float()
            xh = _rows(ln["x"], npix, g["ldy"])[:, :C].float()
            rs = ln["rstd"].reshape(-1)[:npix].float().unsqueeze(1)
            den = C - (1 if ln["unbiased"] else 0)
            part = rs * (gq - gq.mean(dim=1, keepdim=True) - xh * (gq * xh).sum(dim=1, keepdim=True) / den)
            if ln.get("dm") is not None:
                HW, ldm = g["Hout"] * g["Wout"], ln.get("ldm", 0)
                if ldm:
                    torch.as_strided(ln["dm"].reshape(-1), (npix // HW, C), (ldm, 1)).add_(part.view(-1, HW, C).sum(1))
                else:
                    ln["dm"].reshape(-1)[:C] += part.sum(0)
            out = part + (_rows(res, npix, g["ldy"])[:, :C].float() if res is not None else 0)
            _rows(y, npix, g["ldy"])[:, :C] = out.to(TD[dtype])
            return
        ln_backward(tmp, ln["x"], ln.get("m"), res, y, ln.get("dm"), npix, g["Hout"] * g["Wout"], g["Cout"], ln.get("ldm", 0), ln["eps"],
                    ln["unbiased"], dtype)
        return
    B, Hin, Win, Cin, Hout, Wout, Cout, ldy, wrows = (g[k] for k in ("B", "Hin", "Win", "Cin", "Hout", "Wout", "Cout", "ldy", "wrows"))
    taps = 1 if g["mode"] == CONV_1X1 else 9
    T = TD[dtype]
    X = _rows(x, B * Hin * Win, Cin).view(B, Hin, Win, Cin).float().permute(0, 3, 1, 2)
    rows = min(wrows, Cout)
    Wt = w.reshape(-1)[: wrows * taps * Cin].view(wrows, taps, Cin)[:rows].float()
    out = _conv_core(X, Wt, g)
    assert out.shape[2] == Hout and out.shape[3] == Wout, (out.shape, Hout, Wout)
    if bias is not None:
        out = out + bias.reshape(-1)[:rows].float().view(1, -1, 1, 1)
    if rows < Cout:
        out = F.pad(out, (0, 0, 0, 0, 0, Cout - rows))
    if act == ACT_SILU:
        out = F.silu(out)
    if act == ACT_RELU:
        out = F.relu(out)
    npix = B * Hout * Wout
    out = out.permute(0, 2, 3, 1).reshape(npix, Cout).to(T).float()
    if mul is not None:
        mm = _rows(mul, npix, ldy)[:, :Cout].float()
        out = out * (_dsilu(mm) if mulmode == MUL_DSILU else mm)
    if res is not None:
        out = out + _rows(res, npix, ldy)[:, :Cout].float()
    if act == ACT_SILU_PAIR:  # y = silu(a), y2 = silu'(a) with a = the result rounded to the storage type
        a_ = out.to(T).float()
        _rows(y, npix, ldy)[:, :Cout] = F.silu(a_).to(T)
        _rows(y2, npix, ldy)[:, :Cout] = _dsilu(a_).to(T)
        return
    if act == ACT_RELU_PAIR:
        a_ = out.to(T).float()
        _rows(y, npix, ldy)[:, :Cout] = F.relu(a_).to(T)
        _rows(y2, npix, ldy)[:, :Cout] = (a_ > 0).to(T)
        return
    if pool2:  # 2x2 sums of the result as it would have been stored: ((g00 + g01) + g10) + g11 in fp32 (sumpool2's order)
        assert res is None and mul is None and y2 is None and act == ACT_NONE
        gq = out.to(T).float().view(B, Hout, Wout, Cout)
        pooled = ((gq[:, 0::2, 0::2] + gq[:, 0::2, 1::2]) + gq[:, 1::2, 0::2]) + gq[:, 1::2, 1::2]
        _rows(y, npix // 4, ldy)[:, :Cout] = pooled.reshape(npix // 4, Cout).to(T)
        return
    _rows(y, npix, ldy)[:, :Cout] = out.to(T)
    if y2 is not None:
        _rows(y2, npix, ldy)[:, :Cout] = F.silu(out.to(T).float()).to(T)


def new_workspace(device, nbytes=0):
    return None


def conv_wgrad(x, dy, dw, g, dtype, dbias=None, workspace=None):
    B, Hin, Win, Cin, Hout, Wout, Cout, ldy = (g[k] for k in ("B", "Hin", "Win", "Cin", "Hout", "Wout", "Cout", "ldy"))
    taps = 1 if g["mode"] == CONV_1X1 else 9
    X = _rows(x, B * Hin * Win, Cin).view(B, Hin, Win, Cin).float().permute(0, 3, 1, 2)
    with torch.enable_grad():
        Wt = torch.zeros(Cout, taps, Cin, dtype=torch.float32, device=x.device, requires_grad=True)
        out = _conv_core(X, Wt, g)
        gy = _rows(dy, B * Hout * Wout, ldy)[:, :Cout].float().view(B, Hout, Wout, Cout).permute(0, 3, 1, 2)
        (gw,) = torch.autograd.grad(out, Wt, gy)
    dw.reshape(-1)[: Cout * taps * Cin] += gw.reshape(-1)
    if dbias is not None:
        dbias.reshape(-1)[:Cout] += _rows(dy, B * Hout * Wout, ldy)[:, :Cout].float().sum(0)


def conv_wgrad_grouped_supported(g, n, dtype):
    return (conv_patch_supported(g, dtype) or g["mode"] == CONV_1X1) and 2 <= n <= 16  # (the library also excludes the narrow-M form, Cout <= 80)


def conv_wgrad_grouped_workspace_bytes(g, n, dtype):
    return 0


GROUPED_LAUNCHES = []  # (geometry side, layers) of every grouped launch: tests look at how the engine batches them


def conv_wgrad_grouped(items, g, dtype, workspace=None):
    GROUPED_LAUNCHES.append((g["Hout"], len(items)))
    for x, dy, dw, db in items:
        conv_wgrad(x, dy, dw, g, dtype, dbias=db, workspace=workspace)


def _ln(xm, unbiased, eps):
    var, mean = torch.var_mean(xm, dim=-1, keepdim=True, unbiased=bool(unbiased))
    return (xm - mean) / (var + eps).sqrt()


def _mrows(m, npix, HW, C, ldm):
    if m is None:
        return 0.0
    if ldm == 0:
        return m.reshape(-1)[:C].float().view(1, C)
    nb = npix // HW
    mm = torch.as_strided(m.reshape(-1), (nb, C), (ldm, 1))
    return mm.float().repeat_interleave(HW, dim=0)


def ln_forward(x, m, y, npix, HW, C, ldm, eps, unbiased, dtype):
    xm = _rows(x, npix, C).float() + _mrows(m, npix, HW, C, ldm)
    _rows(y, npix, C)[:] = _ln(xm, unbiased, eps).to(TD[dtype])


def ln_backward(dy, x, m, dres, dx, dm, npix, HW, C, ldm, eps, unbiased, dtype):
    with torch.enable_grad():
        xm = (_rows(x, npix, C).float() + _mrows(m, npix, HW, C, ldm)).detach().requires_grad_(True)
        out = _ln(xm, unbiased, eps)
        (g,) = torch.autograd.grad(out, xm, _rows(dy, npix, C).float())
    if dm is not None:
        if ldm == 0:
            dm.reshape(-1)[:C] += g.sum(0)
        else:
            nb = npix // HW
            torch.as_strided(dm.reshape(-1), (nb, C), (ldm, 1)).add_(g.view(nb, HW, C).sum(1))
    if dres is not None:
        g = g + _rows(dres, npix, C).float()
    _rows(dx, npix, C)[:] = g.to(TD[dtype])


def colsum(a, out, rows, C, lda, dtype):
    out.reshape(-1)[:C] += _rows(a, rows, lda)[:, :C].float().sum(0)


def silu(x, y, n, dtype):
    y.reshape(-1)[:n] = F.silu(x.reshape(-1)[:n].float()).to(TD[dtype])


def silu_backward(x, dy, dx, n, dtype):
    dx.reshape(-1)[:n] = (dy.reshape(-1)[:n].float() * _dsilu(x.reshape(-1)[:n].float())).to(TD[dtype])


def sumpool2(g, dx, B, H, W, C, dtype):
    G = _rows(g, B * 2 * H * 2 * W, C).view(B, H, 2, W, 2, C).float().sum(dim=(2, 4))
    _rows(dx, B * H * W, C)[:] = G.reshape(B * H * W, C).to(TD[dtype])


def nchw_to_nhwc(x, eps, musig, y, B, C, HW, ldc, dtype):
    X = x.reshape(-1)[: B * C * HW].view(B, C, HW).float()
    if eps is not None:
        ms = musig.reshape(-1)[: 2 * B].view(B, 2)
        X = ms[:, 0].view(B, 1, 1) * X + ms[:, 1].view(B, 1, 1) * eps.reshape(-1)[: B * C * HW].view(B, C, HW).float()
    Y = _rows(y, B * HW, ldc)
    Y[:] = 0
    Y[:, :C] = X.permute(0, 2, 1).reshape(B * HW, C).to(TD[dtype])


def nhwc_to_nchw(y, out, B, C, HW, ldc, dtype):
    Y = _rows(y, B * HW, ldc)[:, :C].float().view(B, HW, C).permute(0, 2, 1)
    out.reshape(-1)[: B * C * HW] = Y.reshape(-1)


def mse_loss_grad(y, eps, dy, loss_sum, B, C, HW, ldc, gscale, dtype, scaler=None):
    if scaler is not None:  # the kernel multiplies by the float at the pointer: element 0 of a scaler state or of a broadcast gradient
        sc = scaler
        while sc.dim() > 0:
            sc = sc[0]
        gscale = gscale * float(sc)
    Y = _rows(y, B * HW, ldc)[:, :C].float()
    E = eps.reshape(-1)[: B * C * HW].view(B, C, HW).permute(0, 2, 1).reshape(B * HW, C).float()
    d = Y - E
    loss_sum.reshape(-1)[0] += (d * d).sum()
    D = _rows(dy, B * HW, ldc)
    D[:] = 0
    D[:, :C] = (d * gscale).to(TD[dtype])


def sq_err(y, eps, out, loss_sum, B, C, HW, ldc, dtype):
    assert not isinstance(eps, int), "the emulation has no Philox stream: callers inject eps on the CPU"
    Y = _rows(y, B * HW, ldc)[:, :C].float().view(B, HW, C).permute(0, 2, 1)
    sq = (Y - eps.reshape(B, C, HW).float()) ** 2
    out.reshape(-1)[: B * C * HW] = sq.reshape(-1)
    if loss_sum is not None:
        loss_sum.reshape(-1)[0] += sq.sum()
    return True


def timestep_embedding(t, out, n, dim, max_period=10000.0):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    a = t.reshape(-1)[:n, None].float() * freqs[None]
    o = out.reshape(-1)[: n * dim].view(n, dim)
    o[:] = 0
    o[:, : 2 * half] = torch.cat((a.cos(), a.sin()), -1)


def mu_sigma(t, musig, n, eta):
    a = torch.cos(math.acos(math.sqrt(eta)) * t.reshape(-1)[:n].float()) ** 2
    ms = musig.reshape(-1)[: 2 * n].view(n, 2)
    ms[:, 0] = a
    ms[:, 1] = (1 - a * a + eta * eta).sqrt()


def cast_f32(src, dst, n, dtype):
    dst.reshape(-1)[:n] = src.reshape(-1)[:n].to(TD[dtype])


def weight_transpose(w, out, R, NT, K, ldk, ldr, flip, dtype):
    Wm = torch.as_strided(w.reshape(-1), (R, NT, K), (NT * ldk, ldk, 1)).float()
    if flip:
        Wm = Wm.flip(1)
    O = torch.as_strided(out.reshape(-1), (K, NT, R), (NT * ldr, ldr, 1))
    O[:] = Wm.permute(2, 1, 0).to(TD[dtype])


def weight_transpose_batched(flat, out, desc, nconv, dtype):
    for j in range(nconv):
        w_off, o_off, R, NT, K, ldk, ldr, flip = (int(v) for v in desc[j * 8: j * 8 + 8])
        weight_transpose(flat.reshape(-1)[w_off:], out.reshape(-1)[o_off:], R, NT, K, ldk, ldr, flip, dtype)


def adamw_ema(p, g, m, v, ema, shadow, n, lr, beta1, beta2, eps, weight_decay, step, ema_rate, grad_scale, scaler=None):
    P, G, M, V = (a.reshape(-1)[:n] for a in (p, g, m, v))
    if scaler is not None:  # torch.cuda.amp.GradScaler: unscale, skip the step on inf/nan (the EMA still moves)
        grad_scale = grad_scale / float(scaler[0])
        step = int(scaler[3]) + 1
        if float(scaler[2]) != 0:
            if ema is not None:
                ema.reshape(-1)[:n].mul_(ema_rate).add_(P, alpha=1 - ema_rate)
            return
    gi = G * grad_scale
    P.mul_(1 - lr * weight_decay)
    M.mul_(beta1).add_(gi, alpha=1 - beta1)
    V.mul_(beta2).addcmul_(gi, gi, value=1 - beta2)
    bc1 = 1 - beta1**step
    bc2 = 1 - beta2**step
    P.addcdiv_(M, V.sqrt() / math.sqrt(bc2) + eps, value=-lr / bc1)
    if ema is not None:
        ema.reshape(-1)[:n].mul_(ema_rate).add_(P, alpha=1 - ema_rate)
    if shadow is not None:
        shadow.reshape(-1)[:n] = P.to(shadow.dtype)


def grad_scaler_init(state, init_scale):
    state.reshape(-1)[:4] = torch.tensor([init_scale, 0.0, 0.0, 0.0], dtype=torch.float32, device=state.device)


def grad_scaler_check(g, n, state):
    if not bool(torch.isfinite(g.reshape(-1)[:n]).all()):
        state[2] = 1.0


def grad_scaler_update(state, growth, backoff, interval):
    if float(state[2]) != 0:
        state[0] *= backoff
        state[1] = 0.0
    else:
        state[3] += 1
        if float(state[1]) + 1 >= interval:
            state[0] *= growth
            state[1] = 0.0
        else:
            state[1] += 1
    state[2] = 0.0


def ema_update(ema, p, n, rate):
    ema.reshape(-1)[:n].mul_(rate).add_(p.reshape(-1)[:n], alpha=1 - rate)


def _attn(qkv, B, T, C):
    q, k, v = _rows(qkv, B * T, 3 * C).view(B, T, 3 * C).split(C, dim=-1)
    s = torch.einsum("btc,bsc->bts", q, k) / math.sqrt(C)
    lse = torch.logsumexp(s, dim=-1)
    o = torch.einsum("bts,bsc->btc", torch.softmax(s, dim=-1), v)
    return o, lse


def attention_forward(qkv, o, lse, B, T, C, dtype):
    oo, l = _attn(qkv.float(), B, T, C)
    _rows(o, B * T, C)[:] = oo.reshape(B * T, C).to(TD[dtype])
    if lse is not None:
        lse.reshape(-1)[: B * T] = l.reshape(-1)


def attention_backward(qkv, o, d_o, lse, delta_ws, dqkv, B, T, C, dtype):
    with torch.enable_grad():
        q = _rows(qkv, B * T, 3 * C).float().detach().clone().requires_grad_(True)
        oo, _ = _attn(q, B, T, C)
        (g,) = torch.autograd.grad(oo, q, _rows(d_o, B * T, C).float().view(B, T, C))
    _rows(dqkv, B * T, 3 * C)[:] = g.to(TD[dtype])


def window_gather(x, y, nw, F, HW, k, i0, ldc, dtype):
    w = 2 * k + 1
    X = x.reshape(-1)
    Y = _rows(y, nw * HW, ldc)
    Y[:] = 0
    for j in range(nw):
        win = X[(i0 + j) * F * HW: (i0 + j + w) * F * HW].view(w * F, HW)
        Y[j * HW:(j + 1) * HW, : w * F] = win.t().to(TD[dtype])


def window_scatter(y, eps, nw, F, HW, k, i0, nwin_total, ldc, dtype):
    w = 2 * k + 1
    Y = _rows(y, nw * HW, ldc).float().view(nw, HW, ldc)
    Ev = eps.reshape(-1)
    for j in range(nw):
        gi = i0 + j
        for tau in range(w):
            if tau == k or (gi == 0 and tau < k) or (gi == nwin_total - 1 and tau > k):
                Ev[(gi + tau) * F * HW:(gi + tau + 1) * F * HW] = Y[j, :, tau * F:(tau + 1) * F].t().reshape(-1)


def gemv_f32(x, W, bias, y, rows, K, ldk, act=0):
    z = W.reshape(-1)[: rows * ldk].view(rows, ldk)[:, :K].float() @ x.reshape(-1)[:K].float()
    if bias is not None:
        z = z + bias.reshape(-1)[:rows]
    y.reshape(-1)[:rows] = F.silu(z) if act == ACT_SILU else (F.relu(z) if act == ACT_RELU else z)


def conv_center_supported(H, W, Cin, nr, dtype):
    # the emulation has no tile shapes; it mirrors the kernel's domain except the element type (CPU tests run fp32)
    return H % 8 == 0 and W % 16 == 0 and Cin in (64, 128) and 1 <= nr <= 16


def conv_center(x, w, bias, out, B, H, W, Cin, wrows, r0, nr, ostride, dtype):
    """include/c2w_hip.h::c2w_conv_center: rows r0 .. r0 + nr - 1 of the 3x3 convolution, rounded through the compute type, as fp32 planes."""
    X = _rows(x, B * H * W, Cin).float().view(B, H, W, Cin).permute(0, 3, 1, 2)
    Wm = w.reshape(-1)[: wrows * 9 * Cin].view(wrows, 3, 3, Cin)[r0:r0 + nr].float().permute(0, 3, 1, 2)
    bv = bias.reshape(-1)[r0:r0 + nr].float() if bias is not None else None
    Y = F.conv2d(X, Wm, bv, padding=1).to(x.dtype).float()  # (B, nr, H, W)
    O = out.reshape(-1)
    for b in range(B):
        O[b * ostride: b * ostride + nr * H * W] = Y[b].reshape(-1)


def sampler_predict(x, eps, nan_flag, n, a, b):
    X = x.reshape(-1)[:n]
    X.mul_(a).add_(eps.reshape(-1)[:n], alpha=b)
    if nan_flag is not None and not torch.isfinite(X).all():
        nan_flag.reshape(-1)[0] |= 1


def sumsq(v, out, n):
    out.reshape(-1)[0] += v.reshape(-1)[:n].float().square().sum()


def sampler_correct(x, eps, z, sumsq_buf, nan_flag, n, tau, sigma_next):
    delta = tau / (sumsq_buf.reshape(-1)[0] / n)
    X = x.reshape(-1)[:n]
    X.sub_((delta * eps.reshape(-1)[:n] + torch.sqrt(2 * delta) * z.reshape(-1)[:n]) * sigma_next)
    if nan_flag is not None and not torch.isfinite(X).all():
        nan_flag.reshape(-1)[0] |= 1


def guidance(x, eps, yobs, stdv, nobs, F, H, W, s_step, t_step, mu, sigma, gamma):
    L = x.numel() // (F * H * W)
    X = x.reshape(-1)[: L * F * H * W].view(L, F, H, W)
    Ev = eps.reshape(-1)[: L * F * H * W].view(L, F, H, W)
    x0 = (X[::t_step][:nobs] - sigma * Ev[::t_step][:nobs]) / mu
    err = yobs.reshape(nobs, F, H // s_step, W // s_step) - torch.nn.functional.avg_pool2d(x0, s_step)
    gam = gamma.reshape(1, F, 1, 1) if isinstance(gamma, torch.Tensor) else gamma
    var = stdv.reshape(1, F, 1, 1) ** 2 + gam * (sigma / mu) ** 2
    g = (err / var).repeat_interleave(s_step, 2).repeat_interleave(s_step, 3) / (s_step * s_step)
    Ev[::t_step][:nobs] -= sigma * g / mu


def pool_stride(x, y, nobs, F, H, W, s_step, t_step):
    L = x.numel() // (F * H * W)
    X = x.reshape(-1)[: L * F * H * W].view(L, F, H, W)
    y.reshape(-1)[: nobs * F * (H // s_step) * (W // s_step)] = torch.nn.functional.avg_pool2d(X[::t_step][:nobs], s_step).reshape(-1)


def affine_channels(x, y, scale, shift, planes, F, HW):
    X = x.reshape(-1)[: planes * HW].view(planes // F, F, HW)
    y.reshape(-1)[: planes * HW] = (X * scale.reshape(1, F, 1) + shift.reshape(1, F, 1)).reshape(-1)


ALL = [n for n, f in list(globals().items()) if callable(f) and not n.startswith("_") and n not in ("F",)]


def install(monkeypatch, target):
    """Replace every launcher of `target` (climate2weather_amd.ops) with its PyTorch re-statement."""
    import sys
    me = sys.modules[__name__]
    for name in ALL:
        if hasattr(target, name) and name not in ("install",):
            monkeypatch.setattr(target, name, getattr(me, name))
    monkeypatch.setattr(target, "EMULATED", True)
