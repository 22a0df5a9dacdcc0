"""Tensor-level launchers: one Python function per C-ABI entry point of libc2w_hip.so.

Every function takes torch tensors that already live on the GPU (torch is the allocator and the stream
provider, nothing more), hands their device pointers to the HIP library and enqueues on torch's current
stream.  There is no fallback: a missing library or a CPU tensor raises.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import (ACT_NONE, ACT_RELU, ACT_RELU_PAIR, ACT_SILU, ACT_SILU_PAIR, CONV_1X1, CONV_S1, CONV_S2, CONV_TS2, CONV_UP, DTYPE_BF16, DTYPE_F16, DTYPE_F32, MUL_DSILU,  # noqa: F401
                   MUL_PLAIN, ConvArgs, check)

TORCH_DTYPE = {DTYPE_F32: torch.float32, DTYPE_BF16: torch.bfloat16, DTYPE_F16: torch.float16}
ESZ = {DTYPE_F32: 4, DTYPE_BF16: 2, DTYPE_F16: 2}
CK = {DTYPE_F32: 32, DTYPE_BF16: 64, DTYPE_F16: 64}  # channels per 128-byte K chunk


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.C2wError("climate2weather_amd kernels need GPU tensors (no CPU fallback in the product path)")
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    # the raw handle of torch's current stream on the current device: 0.5 us, against 7 us through torch.cuda.current_stream() (a
    # Stream object per call) -- 640 launches per training step, and every one of the first launches of a step sits on the host's
    # critical path when the caller synchronises once per step (training_loop.py:385)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _conv_args(x, w, bias, res, mul, y, g, act, mulmode, y2=None, ln=None, lnf=None, resn=None) -> ConvArgs:
    a = ConvArgs(_p(x), _p(w), _p(bias), _p(res), _p(mul), _p(y), _p(y2), g["B"], g["Hin"], g["Win"], g["Cin"], g["Hout"], g["Wout"],
                 g["Cout"], g["ldy"], g["wrows"], g["mode"], act, mulmode)
    if ln is not None:
        a.ln_x, a.ln_m, a.ln_dm = _p(ln["x"]), _p(ln.get("m")), _p(ln.get("dm"))
        a.ln_ldm, a.ln_unbiased, a.ln_eps = int(ln.get("ldm", 0)), int(ln["unbiased"]), float(ln["eps"])
        a.ln_rstd = _p(ln.get("rstd"))  # with it, ln["x"] holds the normalised rows the forward kept (c2w_hip.h)
    if lnf is not None:
        a.lnf_y, a.lnf_m = _p(lnf["y"]), _p(lnf.get("m"))
        a.lnf_rstd = _p(lnf.get("rstd"))
        a.lnf_mean = _p(lnf.get("mean"))
        a.ln_ldm, a.ln_unbiased, a.ln_eps = int(lnf.get("ldm", 0)), int(lnf["unbiased"]), float(lnf["eps"])
    if resn is not None:  # `res` holds normalised rows: the residual is rebuilt as res / rstd + mean - m (c2w_hip.h, round 6)
        a.res_rstd, a.res_mean, a.res_m = _p(resn["rstd"]), _p(resn["mean"]), _p(resn.get("m"))
    return a


def conv(x, w, bias, y, g: dict, dtype: int, act: int = ACT_NONE, res=None, mul=None, mulmode: int = MUL_PLAIN, naive=False, y2=None,
         ln=None, lnf=None, pool2: bool = False, kvalid: int = 0, wpacked: bool = False, loss: Optional[dict] = None,
         resn: Optional[dict] = None, no_y: bool = False, splitk: Optional[tuple] = None):
    """c2w_conv_forward.  g: geometry dict(B,Hin,Win,Cin,Hout,Wout,Cout,ldy,wrows,mode).
    ln = dict(x, m, dm, ldm, eps, unbiased[, rstd]): fuse the LayerNorm backward into the epilogue (y = res + dLN(conv; x + m),
    dm accumulated) -- only where conv_lnbwd_supported(g, dtype) says so.  With ``rstd`` (what the forward's lnf kept), ``x`` holds the
    NORMALISED rows and ``m`` is not read.
    lnf = dict(y, m, ldm, eps, unbiased[, rstd]): also write y = LN(result + m), the consumer block's normalised input (and, with
    ``rstd``, every pixel row's 1/sigma) -- only where conv_lnfwd_supported(g, dtype) says so."""
    a = _conv_args(x, w, bias, res, mul, y, g, act, mulmode, y2, ln, lnf, resn)
    if no_y:  # with lnf: the result itself is not written (``y``: any valid pointer) -- only where conv_lnfwd_chain_supported says so; so are
        a.flags |= _lib.CONV_NO_Y  # lnf["mean"] and ``resn`` = dict(rstd, mean[, m]): ``res`` holds normalised rows, the residual is rebuilt
    if pool2:  # y: [B][Hout/2][Wout/2][ldy] <- 2x2 sums of the result (only where conv_pool2_supported says so)
        a.flags |= _lib.CONV_POOL2
    if wpacked:  # w: the stage-major copy made by pack_conv_weights_batched (only where conv_wpacked_supported says so)
        a.flags |= _lib.CONV_WPACKED
    a.kvalid = int(kvalid)  # promise: input channels >= kvalid are all zero in x or in w (0: no promise)
    if splitk is not None:  # (scratch tensor, workgroups per tile): exactly conv_splitk_plan's answer for this launch
        ws, ns = splitk
        a.splitk_ws, a.splitk_ws_bytes, a.splitk = _p(ws), ws.numel() * ws.element_size(), int(ns)
    if loss is not None:  # dict(sum, eps, lde, gscale, C[, scaler]): y receives (result - eps rows) * gscale, sum += sum of squares -- only
        a.loss_sum, a.loss_scaler = _p(loss["sum"]), _p(loss.get("scaler"))  # where conv_loss_supported(g, dtype) says so
        a.loss_eps, a.loss_lde, a.loss_gscale, a.loss_C = _p(loss["eps"]), int(loss["lde"]), float(loss["gscale"]), int(loss["C"])
    check(_lib.load().c2w_conv_forward(ctypes.byref(a), dtype, int(naive), _stream()), "c2w_conv_forward")  # naive: 0 product, 1 direct, 2 gather


def conv_wpacked_supported(g: dict, dtype: int) -> bool:
    """True when c2w_conv_forward takes this geometry with stage-major packed weights (it goes to the 16x16-tile kernel)."""
    a = ConvArgs(None, None, None, None, None, None, None, g["B"], g["Hin"], g["Win"], g["Cin"], g["Hout"], g["Wout"], g["Cout"], g["ldy"],
                 g["wrows"], g["mode"], ACT_NONE, MUL_PLAIN)
    return bool(_lib.load().c2w_conv_wpacked_supported(ctypes.byref(a), dtype))


def packed_conv_weights_numel(rows: int, cin: int) -> int:
    """Elements of one stage-major packed 3x3 weight matrix (rows padded to 128)."""
    return 9 * cin * ((rows + 127) // 128 * 128)


def pack_conv_weights_batched(src, dst, desc, n: int, dtype: int) -> None:
    """c2w_pack_conv_weights_batched: desc = int64 tensor [n][4] = (src_off, dst_off, rows, cin), offsets in elements."""
    check(_lib.load().c2w_pack_conv_weights_batched(_p(src), _p(dst), _p(desc), n, dtype, _stream()), "c2w_pack_conv_weights_batched")


EMULATED = False  # tests/emu_ops.install sets it: the launchers below are PyTorch restatements running on CPU tensors
KNOBS_GENERATION = 0  # bumped by knobs_reload(): callers that memoise dispatch answers (engine._pk_ok) key them on it


def knobs_reload() -> None:
    """Re-read the library's run-time knobs (C2W_* dispatch overrides) from the environment: it reads them once, at load."""
    global KNOBS_GENERATION
    _lib.load().c2w_knobs_reload()  # (load() re-applies the host's knob defaults: a test that deleted a variable gets the HOST default back)
    KNOBS_GENERATION += 1


def conv_patch_supported(g: dict, dtype: int) -> bool:
    a = ConvArgs(None, None, None, None, None, None, None, g["B"], g["Hin"], g["Win"], g["Cin"], g["Hout"], g["Wout"], g["Cout"], g["ldy"],
                 g["wrows"], g["mode"], ACT_NONE, MUL_PLAIN)
    return bool(_lib.load().c2w_conv_patch_supported(ctypes.byref(a), dtype))


def conv_pool2_supported(g: dict, dtype: int) -> bool:
    a = ConvArgs(None, None, None, None, None, None, None, g["B"], g["Hin"], g["Win"], g["Cin"], g["Hout"], g["Wout"], g["Cout"], g["ldy"],
                 g["wrows"], g["mode"], ACT_NONE, MUL_PLAIN)
    return bool(_lib.load().c2w_conv_pool2_supported(ctypes.byref(a), dtype))


def conv_lnfwd_supported(g: dict, dtype: int) -> bool:
    a = ConvArgs(None, None, None, None, None, None, None, g["B"], g["Hin"], g["Win"], g["Cin"], g["Hout"], g["Wout"], g["Cout"], g["ldy"],
                 g["wrows"], g["mode"], ACT_NONE, MUL_PLAIN)
    return bool(_lib.load().c2w_conv_lnfwd_supported(ctypes.byref(a), dtype))


def conv_splitk_plan(g: dict, dtype: int, act: int = ACT_NONE) -> tuple:
    """(workgroups per output tile, scratch bytes) for a conv launch with bias / activation / mul / res epilogues only
    (include/c2w_hip.h::c2w_conv_splitk_plan); (1, 0): the launch does not split."""
    a = _geom_args(g)
    a.act = act
    nbytes = ctypes.c_ulonglong(0)
    ns = int(_lib.load().c2w_conv_splitk_plan(ctypes.byref(a), dtype, ctypes.byref(nbytes)))
    if ns < 0:
        check(ns, "c2w_conv_splitk_plan")
    return ns, int(nbytes.value)


def conv_lnfwd_chain_supported(g: dict, dtype: int) -> bool:
    """True when c2w_conv_forward takes lnf["mean"] / resn / no_y for this geometry (include/c2w_hip.h)."""
    return bool(_lib.load().c2w_conv_lnfwd_chain_supported(ctypes.byref(_geom_args(g)), dtype))


def conv_loss_supported(g: dict, dtype: int) -> bool:
    """True when c2w_conv_forward can fuse the training loss into this launch (include/c2w_hip.h::c2w_conv_loss_supported)."""
    return bool(_lib.load().c2w_conv_loss_supported(ctypes.byref(_geom_args(g)), dtype))


def conv_lnbwd_supported(g: dict, dtype: int) -> bool:
    a = ConvArgs(None, None, None, None, None, None, None, g["B"], g["Hin"], g["Win"], g["Cin"], g["Hout"], g["Wout"], g["Cout"], g["ldy"],
                 g["wrows"], g["mode"], ACT_NONE, MUL_PLAIN)
    return bool(_lib.load().c2w_conv_lnbwd_supported(ctypes.byref(a), dtype))


def conv_dispatch(g: dict, dtype: int, pool2: bool = False, fused_ln: bool = False) -> int:
    """Kernel family (``_lib.KERNEL_*``) c2w_conv_forward runs this geometry on; ``fused_ln``: with a LayerNorm epilogue requested."""
    a = ConvArgs(None, None, None, None, None, None, None, g["B"], g["Hin"], g["Win"], g["Cin"], g["Hout"], g["Wout"], g["Cout"], g["ldy"],
                 g["wrows"], g["mode"], ACT_NONE, MUL_PLAIN)
    if pool2:
        a.flags = _lib.CONV_POOL2
    if fused_ln:
        a.lnf_y = ctypes.c_void_p(16)  # only tested against NULL
    return int(_lib.load().c2w_conv_dispatch(ctypes.byref(a), dtype))


def conv_wgrad_dispatch(g: dict, dtype: int) -> int:
    a = ConvArgs(None, None, None, None, None, None, None, g["B"], g["Hin"], g["Win"], g["Cin"], g["Hout"], g["Wout"], g["Cout"], g["ldy"],
                 g["wrows"], g["mode"], ACT_NONE, MUL_PLAIN)
    rc = int(_lib.load().c2w_conv_wgrad_dispatch(ctypes.byref(a), dtype))
    if rc < 0:
        check(rc, "c2w_conv_wgrad_dispatch")
    return rc


def conv_wgrad(x, dy, dw, g: dict, dtype: int, dbias=None, workspace: Optional[torch.Tensor] = None):
    """c2w_conv_wgrad: dw += dY^T . gather(x); dbias (optional) += column sums of dY.
    ``workspace``: fp32 scratch tensor for the split-K partial sums (``new_workspace``), handed over per call; one per stream.
    Without it (or when it is too small for the geometry) the partial sums are combined with fp32 atomics."""
    a = _conv_args(x, None, None, None, None, dy, g, 0, 0)
    a.w = None
    nbytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    check(_lib.load().c2w_conv_wgrad(ctypes.byref(a), _p(dw), _p(dbias), _p(workspace), nbytes, dtype, _stream()), "c2w_conv_wgrad")


def conv_wgrad_workspace_bytes(g: dict, dtype: int) -> int:
    a = ConvArgs(None, None, None, None, None, None, None, g["B"], g["Hin"], g["Win"], g["Cin"], g["Hout"], g["Wout"], g["Cout"], g["ldy"],
                 g["wrows"], g["mode"], ACT_NONE, MUL_PLAIN)
    n = int(_lib.load().c2w_conv_wgrad_workspace_bytes(ctypes.byref(a), dtype))
    if n < 0:
        check(n, "c2w_conv_wgrad_workspace_bytes")
    return n


def _geom_args(g: dict):
    return ConvArgs(None, None, None, None, None, None, None, g["B"], g["Hin"], g["Win"], g["Cin"], g["Hout"], g["Wout"], g["Cout"], g["ldy"],
                    g["wrows"], g["mode"], ACT_NONE, MUL_PLAIN)


def conv_wgrad_grouped_supported(g: dict, n: int, dtype: int) -> bool:
    """Do ``n`` weight gradients of geometry ``g`` run as ONE launch (include/c2w_hip.h::c2w_conv_wgrad_grouped)?"""
    return bool(_lib.load().c2w_conv_wgrad_grouped_supported(ctypes.byref(_geom_args(g)), n, dtype))


def conv_wgrad_grouped_workspace_bytes(g: dict, n: int, dtype: int) -> int:
    nb = int(_lib.load().c2w_conv_wgrad_grouped_workspace_bytes(ctypes.byref(_geom_args(g)), n, dtype))
    if nb < 0:
        check(nb, "c2w_conv_wgrad_grouped_workspace_bytes")
    return nb


def conv_wgrad_grouped(items, g: dict, dtype: int, workspace: Optional[torch.Tensor] = None):
    """items: [(x, dy, dw, dbias or None)] of layers that share geometry ``g``: every dw += dY^T . patches(x), every dbias += column sums
    of dY, by one launch (+ one reduction launch when the plan splits K)."""
    arr = (_lib.WgradItem * len(items))()
    for i, (x, dy, dw, db) in enumerate(items):
        arr[i].x, arr[i].dy, arr[i].dw, arr[i].dbias = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), (db.data_ptr() if db is not None else None)
    nbytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    check(_lib.load().c2w_conv_wgrad_grouped(ctypes.byref(_geom_args(g)), arr, len(items), _p(workspace), nbytes, dtype, _stream()),
          "c2w_conv_wgrad_grouped")


# covers every layer of the default network at any batch (75.5 MB per launch, independent of the batch size); C2W_WORKSPACE_MB: A/B runs
# that raise the number of splits (C2W_WGRAD_WGS)
WORKSPACE_BYTES = int(__import__("os").environ.get("C2W_WORKSPACE_MB", "96")) << 20


def new_workspace(device, nbytes: int = WORKSPACE_BYTES) -> torch.Tensor:
    """A scratch buffer for ``conv_wgrad`` (allocated on torch's current stream: keep one per stream that launches weight gradients)."""
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device)


def ln_forward(x, m, y, npix, HW, C, ldm, eps, unbiased, dtype):
    check(_lib.load().c2w_ln_forward(_p(x), _p(m), _p(y), npix, HW, C, ldm, eps, int(unbiased), dtype, _stream()), "c2w_ln_forward")


def ln_backward(dy, x, m, dres, dx, dm, npix, HW, C, ldm, eps, unbiased, dtype):
    check(_lib.load().c2w_ln_backward(_p(dy), _p(x), _p(m), _p(dres), _p(dx), _p(dm), npix, HW, C, ldm, eps, int(unbiased), dtype,
                                      _stream()), "c2w_ln_backward")


def colsum(a, out, rows, C, lda, dtype):
    check(_lib.load().c2w_colsum(_p(a), _p(out), rows, C, lda, dtype, _stream()), "c2w_colsum")


def silu(x, y, n, dtype):
    check(_lib.load().c2w_silu(_p(x), _p(y), n, dtype, _stream()), "c2w_silu")


def silu_backward(x, dy, dx, n, dtype):
    check(_lib.load().c2w_silu_backward(_p(x), _p(dy), _p(dx), n, dtype, _stream()), "c2w_silu_backward")


def sumpool2(g, dx, B, H, W, C, dtype):
    check(_lib.load().c2w_sumpool2(_p(g), _p(dx), B, H, W, C, dtype, _stream()), "c2w_sumpool2")


def upsample2(x, y, B, H, W, C, dtype):
    check(_lib.load().c2w_upsample2(_p(x), _p(y), B, H, W, C, dtype, _stream()), "c2w_upsample2")


def nchw_to_nhwc(x, eps, musig, y, B, C, HW, ldc, dtype):
    check(_lib.load().c2w_nchw_to_nhwc(_p(x), _p(eps), _p(musig), _p(y), B, C, HW, ldc, dtype, _stream()), "c2w_nchw_to_nhwc")


def nhwc_to_nchw(y, out, B, C, HW, ldc, dtype):
    check(_lib.load().c2w_nhwc_to_nchw(_p(y), _p(out), B, C, HW, ldc, dtype, _stream()), "c2w_nhwc_to_nchw")


def mse_loss_grad(y, eps, dy, loss_sum, B, C, HW, ldc, gscale, dtype, scaler=None):
    """``scaler``: the 4-float device state of the dynamic loss scale (fp16 training) or None."""
    check(_lib.load().c2w_mse_loss_grad_scaled(_p(y), _p(eps), _p(dy), _p(loss_sum), B, C, HW, ldc, gscale, _p(scaler), dtype, _stream()),
          "c2w_mse_loss_grad")


def philox_normal(out, n, seed):
    """out[:n] = the N(0,1) stream of ``seed`` (the one the *_noise launchers regenerate)."""
    check(_lib.load().c2w_philox_normal(_p(out), n, int(seed), _stream()), "c2w_philox_normal")


def nchw_to_nhwc_noise(x, seed, musig, y, B, C, HW, ldc, dtype) -> bool:
    """nchw_to_nhwc with eps := philox stream of ``seed``; False if the shape is not supported (caller materialises the stream)."""
    rc = _lib.load().c2w_nchw_to_nhwc_noise(_p(x), int(seed), _p(musig), _p(y), B, C, HW, ldc, dtype, _stream())
    if rc == -3:
        return False
    check(rc, "c2w_nchw_to_nhwc_noise")
    return True


def nchw_to_nhwc_noise_rows(x, img_off, seed, musig, y, erows, B, C, HW, ldc, lde, dtype) -> bool:
    """nchw_to_nhwc_noise (``img_off`` None) / windows_to_nhwc_noise that also writes the noise it mixed in, rounded to half precision,
    as NHWC rows ``erows`` [B*HW][lde] float16 -- and mixes THAT rounded noise (include/c2w_hip.h).  False: shape not supported."""
    rc = _lib.load().c2w_nchw_to_nhwc_noise_rows(_p(x), _p(img_off), int(seed), _p(musig), _p(y), _p(erows), B, C, HW, ldc, lde, dtype, _stream())
    if rc == -3:
        return False
    check(rc, "c2w_nchw_to_nhwc_noise_rows")
    return True


def windows_to_nhwc_noise(data, img_off, seed, musig, y, B, C, HW, ldc, dtype) -> bool:
    """nchw_to_nhwc_noise reading image b from ``data`` at float offset ``img_off[b]`` (int64 device tensor): the training batch is
    never gathered.  False if the shape is not supported (caller gathers and takes the dense path)."""
    rc = _lib.load().c2w_windows_to_nhwc_noise(_p(data), _p(img_off), int(seed), _p(musig), _p(y), B, C, HW, ldc, dtype, _stream())
    if rc == -3:
        return False
    check(rc, "c2w_windows_to_nhwc_noise")
    return True


def mse_loss_grad_noise(y, seed, dy, loss_sum, B, C, HW, ldc, gscale, dtype, scaler=None) -> bool:
    rc = _lib.load().c2w_mse_loss_grad_noise(_p(y), int(seed), _p(dy), _p(loss_sum), B, C, HW, ldc, gscale, _p(scaler), dtype, _stream())
    if rc == -3:
        return False
    check(rc, "c2w_mse_loss_grad_noise")
    return True


def sq_err(y, eps, out, loss_sum, B, C, HW, ldc, dtype) -> bool:
    """out (B,C,H,W) fp32 = (y - eps)^2 from NHWC rows ``y``; loss_sum[0] += its sum; ``eps``: an fp32 (B,C,H,W) tensor, or an int seed
    (the stream the *_noise launchers regenerate).  False if the shape is not supported (caller converts the layout and uses tensor
    arithmetic)."""
    if isinstance(eps, int):
        rc = _lib.load().c2w_sq_err_noise(_p(y), int(eps), _p(out), _p(loss_sum), B, C, HW, ldc, dtype, _stream())
    else:
        rc = _lib.load().c2w_sq_err(_p(y), _p(eps), _p(out), _p(loss_sum), B, C, HW, ldc, dtype, _stream())
    if rc == -3:
        return False
    check(rc, "c2w_sq_err")
    return True


def timestep_embedding(t, out, n, dim, max_period=10000.0):
    check(_lib.load().c2w_timestep_embedding(_p(t), _p(out), n, dim, max_period, _stream()), "c2w_timestep_embedding")


def mu_sigma(t, musig, n, eta):
    check(_lib.load().c2w_mu_sigma(_p(t), _p(musig), n, eta, _stream()), "c2w_mu_sigma")


def conv_center_supported(H: int, W: int, Cin: int, nr: int, dtype: int) -> bool:
    return bool(_lib.load().c2w_conv_center_supported(H, W, Cin, nr, dtype))


def conv_center(x, w, bias, out, B, H, W, Cin, wrows, r0, nr, ostride, dtype):
    """rows r0 .. r0 + nr - 1 of the 3x3 output convolution over the NHWC rows x, as fp32 planes out[b * ostride + c * H * W + pix]
    (include/c2w_hip.h::c2w_conv_center): the frames the sampler's fold keeps, written where it puts them."""
    check(_lib.load().c2w_conv_center(_p(x), _p(w), _p(bias), _p(out), B, H, W, Cin, wrows, r0, nr, ostride, dtype, _stream()), "c2w_conv_center")


def gemv_f32(x, W, bias, y, rows, K, ldk, act=ACT_NONE):
    """y[r] = act(bias[r] + W[r][:K] . x): a Linear layer applied to ONE row (include/c2w_hip.h::c2w_gemv_f32)."""
    check(_lib.load().c2w_gemv_f32(_p(x), _p(W), _p(bias), _p(y), rows, K, ldk, act, _stream()), "c2w_gemv_f32")


def publish_scalar(src, host_slot_ptr: int, seq: int):
    """src: 0-d / 1-element fp32 device tensor; host_slot_ptr: address of two ints of pinned host memory (value bits, sequence number)."""
    check(_lib.load().c2w_publish_scalar(_p(src), ctypes.c_void_p(host_slot_ptr), int(seq), _stream()), "c2w_publish_scalar")


class HostRing:
    """A ring of (value bits, sequence number) slots in pinned host memory that kernels publish 4-byte device scalars into
    (publish_scalar) and the host polls WITHOUT synchronising a stream: the loss value of a training step (Engine.publish), the
    sampler's NaN flag of every step (pipelines.SDAPipeline.sample).  A reader that comes more than ``slots`` publications late
    finds a newer number in its slot and gets None."""

    def __init__(self, slots: int = 64):
        self.slots = slots
        self.buf = torch.zeros((slots, 2), dtype=torch.int32).pin_memory()
        self.arr = self.buf.numpy()
        self.n = 0

    def publish(self, scalar):
        """Enqueue, behind whatever produced the 4-byte device scalar, its copy into the next slot; -> (slot, sequence number)."""
        self.n += 1
        slot = self.n % self.slots
        publish_scalar(scalar, self.buf.data_ptr() + 8 * slot, self.n)
        return slot, self.n

    def read_bits(self, slot: int, seq: int, timeout_s: float = 20.0):
        """The published 32 bits as an int once the device has written publication ``seq`` (polling host memory), or None if the slot has
        been reused or nothing arrived within ``timeout_s`` (0: do not wait)."""
        import time
        t0 = None
        while True:
            got = int(self.arr[slot, 1])
            if got == seq:
                return int(self.arr[slot, 0])
            if got > seq or timeout_s <= 0:
                return None
            if t0 is None:
                t0 = time.perf_counter()
            elif time.perf_counter() - t0 > timeout_s:
                return None
            time.sleep(0)


def cast_f32(src, dst, n, dtype):
    check(_lib.load().c2w_cast_f32(_p(src), _p(dst), n, dtype, _stream()), "c2w_cast_f32")


def weight_transpose(w, out, R, NT, K, ldk, ldr, flip, dtype):
    check(_lib.load().c2w_weight_transpose(_p(w), _p(out), R, NT, K, ldk, ldr, int(flip), dtype, _stream()), "c2w_weight_transpose")


def weight_transpose_batched(flat, out, desc, nconv, dtype):
    check(_lib.load().c2w_weight_transpose_batched(_p(flat), _p(out), _p(desc), nconv, dtype, _stream()), "c2w_weight_transpose_batched")


def adamw_ema(p, g, m, v, ema, shadow, n, lr, beta1, beta2, eps, weight_decay, step, ema_rate, grad_scale, scaler=None):
    """``shadow``: None or the 16-bit weight copy (bf16 / fp16 tensor) to refresh; ``scaler``: loss-scale state or None."""
    sdt = DTYPE_F16 if shadow is not None and shadow.dtype == torch.float16 else DTYPE_BF16
    check(_lib.load().c2w_adamw_ema_scaled(_p(p), _p(g), _p(m), _p(v), _p(ema), _p(shadow), sdt, n, lr, beta1, beta2, eps, weight_decay,
                                           step, ema_rate, grad_scale, _p(scaler), _stream()), "c2w_adamw_ema")


def grad_scaler_init(state, init_scale):
    check(_lib.load().c2w_grad_scaler_init(_p(state), init_scale, _stream()), "c2w_grad_scaler_init")


def grad_scaler_check(g, n, state):
    check(_lib.load().c2w_grad_scaler_check(_p(g), n, _p(state), _stream()), "c2w_grad_scaler_check")


def grad_scaler_update(state, growth, backoff, interval):
    check(_lib.load().c2w_grad_scaler_update(_p(state), growth, backoff, interval, _stream()), "c2w_grad_scaler_update")


def ema_update(ema, p, n, rate):
    check(_lib.load().c2w_ema_update(_p(ema), _p(p), n, rate, _stream()), "c2w_ema_update")


def attention_forward(qkv, o, lse, B, T, C, dtype):
    check(_lib.load().c2w_attention_forward(_p(qkv), _p(o), _p(lse), B, T, C, dtype, _stream()), "c2w_attention_forward")


def attention_backward(qkv, o, d_o, lse, delta_ws, dqkv, B, T, C, dtype):
    check(_lib.load().c2w_attention_backward(_p(qkv), _p(o), _p(d_o), _p(lse), _p(delta_ws), _p(dqkv), B, T, C, dtype, _stream()),
          "c2w_attention_backward")


def window_gather(x, y, nw, F, HW, k, i0, ldc, dtype):
    check(_lib.load().c2w_window_gather(_p(x), _p(y), nw, F, HW, k, i0, ldc, dtype, _stream()), "c2w_window_gather")


def window_scatter(y, eps, nw, F, HW, k, i0, nwin_total, ldc, dtype):
    check(_lib.load().c2w_window_scatter(_p(y), _p(eps), nw, F, HW, k, i0, nwin_total, ldc, dtype, _stream()), "c2w_window_scatter")


def sampler_predict(x, eps, nan_flag, n, a, b):
    check(_lib.load().c2w_sampler_predict(_p(x), _p(eps), _p(nan_flag), n, a, b, _stream()), "c2w_sampler_predict")


def sumsq(v, out, n):
    check(_lib.load().c2w_sumsq(_p(v), _p(out), n, _stream()), "c2w_sumsq")


def sampler_correct(x, eps, z, sumsq_buf, nan_flag, n, tau, sigma_next):
    check(_lib.load().c2w_sampler_correct(_p(x), _p(eps), _p(z), _p(sumsq_buf), _p(nan_flag), n, tau, sigma_next, _stream()),
          "c2w_sampler_correct")


def guidance(x, eps, yobs, stdv, nobs, F, H, W, s_step, t_step, mu, sigma, gamma):
    """``gamma``: a float, or a device tensor of F values (one per variable, exp/downscaling.py:228-233)."""
    if isinstance(gamma, torch.Tensor):
        check(_lib.load().c2w_guidance_per_variable(_p(x), _p(eps), _p(yobs), _p(stdv), _p(gamma), nobs, F, H, W, s_step, t_step, mu, sigma,
                                                    _stream()), "c2w_guidance_per_variable")
        return
    check(_lib.load().c2w_guidance(_p(x), _p(eps), _p(yobs), _p(stdv), nobs, F, H, W, s_step, t_step, mu, sigma, gamma, _stream()),
          "c2w_guidance")


def pool_stride(x, y, nobs, F, H, W, s_step, t_step):
    check(_lib.load().c2w_pool_stride(_p(x), _p(y), nobs, F, H, W, s_step, t_step, _stream()), "c2w_pool_stride")


def affine_channels(x, y, scale, shift, planes, F, HW):
    check(_lib.load().c2w_affine_channels(_p(x), _p(y), _p(scale), _p(shift), planes, F, HW, _stream()), "c2w_affine_channels")
