"""Host-side engine: lays the ScoreUNet parameters out for the HIP kernels and sequences the launches of one
forward (and, for training / exact guidance, the hand-written backward) of the reference's hot path

    ScoreUNet.forward  (model/score.py:59-70)  ->  UNet.forward  (model/nn.py:220-242)
    ModResidualBlock   (model/nn.py:27-28)         AttentionBlock (model/nn.py:49-59)

Data layout in HBM
  * parameters: ONE flat fp32 buffer; every nn.Parameter is a strided view into it.  Conv2d weights are stored
    [Cout][kh][kw][Cin] (a channels_last view of the reference's OIHW tensor) which is exactly the K-contiguous
    operand the implicit GEMM wants; the 30 modulation Linears sit back to back so that one GEMM produces every
    block's modulation vector.  The 16-bit modes (bf16, fp16) keep a shadow of the same buffer in their format (same offsets).
  * gradients: one flat fp32 buffer with the same offsets (a single RCCL all-reduce covers all 228 tensors).
  * activations: NHWC rows [B*H*W][C] in the compute dtype (fp32 or bf16); the reference's NCHW fp32 tensors only
    exist at the module boundary.
Everything runs on torch's current stream; torch is the allocator, nothing else.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import ops
from .nn import BlockSpec, LevelSpec
from .ops import (ACT_NONE, ACT_RELU, ACT_RELU_PAIR, ACT_SILU, ACT_SILU_PAIR, CONV_1X1, CONV_S1, CONV_S2, CONV_TS2, CONV_UP, DTYPE_BF16, DTYPE_F16, DTYPE_F32, MUL_DSILU, MUL_PLAIN, TORCH_DTYPE)

import weakref

LN_EPS = 1e-5
_ENGINE_BY_PARAM_ID: Dict[int, "weakref.ReferenceType"] = {}  # id(Parameter) -> weakref(Engine); nothing is stored ON the Parameter (they are pickled / deep-copied)


def engine_of_parameter(p):
    """The Engine whose flat buffer the Parameter OBJECT ``p`` is bound to, or None (optim.AdamW finds its engine through this).
    An engine keeps its Parameters alive, so an id cannot be recycled while its entry is valid; the identity check covers the rest."""
    ref = _ENGINE_BY_PARAM_ID.get(id(p))
    eng = ref() if ref is not None else None
    if eng is None:
        _ENGINE_BY_PARAM_ID.pop(id(p), None)
        return None
    return eng if eng._bound_by_id.get(id(p)) is p else None


ALIGN = 64  # elements; keeps every region 256-B aligned in fp32 and 128-B aligned in the bf16 shadow


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


@dataclass
class ConvRec:
    """One weight matrix [rows][taps][cin] (+ bias[rows]) inside the flat buffer -- dense, so that every Parameter is a
    non-overlapping dense view (optimizers, DDP bucket views and the autograd layout contract all assume that)."""
    name: str
    rows: int  # Cout
    cin: int  # real input channels
    kstride: int  # channel stride of the forward operand (>= cin, padded for the network input)
    taps: int
    w_off: int = 0
    b_off: int = 0
    lin: bool = False  # fp32-only GEMM (time MLP, modulation projections)
    dg_ld: int = 0  # K stride of the input-gradient operand = channel stride of this layer's output rows
    dg_off: int = -1  # offset inside the dgrad weight buffer (-1: not needed)
    flip: bool = False


class Layout:
    """Flat parameter layout derived from the module tree."""

    def __init__(self, net):
        unet = net.unet
        self.levels: List[LevelSpec] = unet.spec()
        self.E = unet.mod_features
        self.in_channels, self.out_channels = unet.in_channels, unet.out_channels
        self.cin_pad = _round_up(unet.in_channels, 64)
        self.cout_pad = _round_up(unet.out_channels, 64)
        self.noise_features = net.noise_features
        self.activation = getattr(unet, "activation_kind", "silu")  # of the residual blocks (model/nn.py:156); the time MLP is always SiLU
        self.convs: Dict[str, ConvRec] = {}
        self.views: Dict[str, Tuple[int, Tuple[int, ...], Tuple[int, ...]]] = {}  # param name -> (offset, shape, strides)
        off = 0
        self._off = 0
        # Flat order = reverse of the order in which backward finalises gradients (time MLP last, output conv first), so
        # the finished part of the gradient buffer is always a growing suffix -> bucketed all-reduce can chase it.
        self.forcing_dim = net.map_forcing.in_features if getattr(net, "map_forcing", None) is not None else 0
        if self.forcing_dim:  # model/score.py:49-51; its gradient is final together with map_layer1's: same place in the flat order
            self._add("map_forcing", self.E, self.forcing_dim, 1, kstride=_round_up(self.forcing_dim, 32), lin=True)
        self._add("map_layer0", self.E, self.noise_features, 1, lin=True)
        self._add("map_layer1", self.E, self.E, 1, lin=True)
        off = self._off
        # modulation projections, back to back in execution order:  Wp_all [sumC][E], bp_all [sumC]
        blocks: List[BlockSpec] = []
        for lv in self.levels:
            blocks += [b for b in lv.descent if b.kind == "res"]
        for lv in reversed(self.levels):
            blocks += [b for b in lv.ascent if b.kind == "res"]
        mo = 0
        for b in blocks:
            b.mod_offset = mo
            mo += b.channels
        self.sum_c = mo
        self.proj_w_off = off
        for b in blocks:
            self.views[f"unet.{b.key}.project.0.weight"] = (off, (b.channels, self.E), (self.E, 1))
            off += b.channels * self.E
        off = _round_up(off, ALIGN)
        self.proj_b_off = off
        for b in blocks:
            self.views[f"unet.{b.key}.project.0.bias"] = (off, (b.channels,), (1,))
            off += b.channels
        off = _round_up(off, ALIGN)
        self.convs["proj"] = ConvRec("proj", self.sum_c, self.E, self.E, 1, self.proj_w_off, self.proj_b_off, lin=True, dg_ld=self.sum_c)

        self._off = off
        add = self._add

        L = len(self.levels)
        for i, lv in enumerate(self.levels):
            if i == 0:
                add("unet." + lv.head_key, lv.channels, self.in_channels, 9, kstride=self.cin_pad, dg_ld=lv.channels, flip=True)
            else:
                add("unet." + lv.head_key, lv.channels, self.levels[i - 1].channels, 9, flip=False)
            for b in lv.descent:
                self._add_block(add, b)
        for i in reversed(range(L)):
            lv = self.levels[i]
            for b in lv.ascent:
                self._add_block(add, b)
            if i > 0:
                add("unet." + lv.tail_key, self.levels[i - 1].channels, lv.channels, 9, flip=True)
            else:
                add("unet." + lv.tail_key, self.out_channels, lv.channels, 9, dg_ld=self.cout_pad, flip=True)
        self.numel = self._off
        # dgrad operand buffers: [cin][taps][dg_ld]
        dg = 0
        dgl = 0
        for rec in self.convs.values():
            if rec.name in ("map_layer0", "map_forcing"):
                continue
            size = _round_up(rec.cin * rec.taps * rec.dg_ld, ALIGN)
            if rec.lin:
                rec.dg_off = dgl
                dgl += size
            else:
                rec.dg_off = dg
                dg += size
        self.dg_numel, self.dg_lin_numel = dg, dgl

    def _add(self, name: str, rows: int, cin: int, taps: int, kstride: Optional[int] = None, lin=False, dg_ld=None, flip=False, ndim=4):
        off = self._off
        rec = ConvRec(name, rows, cin, kstride or cin, taps, lin=lin, dg_ld=dg_ld if dg_ld is not None else rows, flip=flip)
        rec.w_off = off
        ks = cin  # storage is dense; a layer whose operand stride differs (kstride > cin) reads a padded copy (Engine._w)
        if taps == 9:
            self.views[name + ".weight"] = (off, (rows, cin, 3, 3), (9 * ks, 1, 3 * ks, ks))
        elif ndim == 3:
            self.views[name + ".weight"] = (off, (rows, cin, 1), (ks, 1, 1))
        else:
            self.views[name + ".weight"] = (off, (rows, cin), (ks, 1))
        off = _round_up(off + rows * taps * ks, ALIGN)
        rec.b_off = off
        self.views[name + ".bias"] = (off, (rows,), (1,))
        self._off = _round_up(off + rows, ALIGN)
        self.convs[name] = rec
        return rec

    @staticmethod
    def _add_block(add, b: BlockSpec):
        p = "unet." + b.key
        if b.kind == "res":
            add(p + ".residue.1", b.channels, b.channels, 9, flip=True)
            add(p + ".residue.3", b.channels, b.channels, 9, flip=True)
        else:
            add(p + ".qkv", 3 * b.channels, b.channels, 1, ndim=3)
            add(p + ".proj_out", b.channels, b.channels, 1, ndim=3)


class Tape:
    """Backward closures recorded by a training forward (applied in reverse)."""

    def __init__(self):
        self.steps: List[Callable] = []
        self.gskip: Dict[int, torch.Tensor] = {}
        self.meta: dict = {}
        # called with the lowest flat offset whose gradient is final AND whose weights the rest of this backward no longer reads from
        # the flat parameter buffer (input-gradient operands are copies made before): the trainer may all-reduce and UPDATE that suffix
        self.progress: Optional[Callable[[int], None]] = None

    def done(self, off: int) -> None:
        if self.progress is not None:
            self.progress(off)


class Engine:
    LINEAR_DGRAD_SPLIT_MIN_ROWS = 2048  # Linear layers at least this wide take the split-reduction input-gradient route

    def __init__(self, net):
        self.layout = Layout(net)
        self.ln_unbiased = bool(getattr(net, "ln_unbiased", True))
        self.flat: Optional[torch.Tensor] = None
        self.flat_grad: Optional[torch.Tensor] = None
        self.shadows: Dict[int, torch.Tensor] = {}   # 16-bit dtype -> copy of the flat buffer in that format
        self._shadow_ver: Dict[int, object] = {}
        self.dg: Dict[int, torch.Tensor] = {}
        self.dg_lin: Optional[torch.Tensor] = None
        self._dg_ver: Dict[object, int] = {}
        self._dg_desc: Dict[object, tuple] = {}
        self._wpad: Dict[object, tuple] = {}   # (name, dtype) -> (version, zero-padded operand copy)
        # stage-major packed copies of the 3x3 weights for the 16x16-tile kernel (ops.pack_conv_weights_batched): "f" forward operands
        # (from the 16-bit shadow), "d" input-gradient operands (from the transposed copies); (kind, dtype) -> buffer / version / table
        self._pk: Dict[tuple, torch.Tensor] = {}
        self._pk_tab: Dict[tuple, tuple] = {}
        self._pk_ok: Dict[tuple, bool] = {}    # launch geometry -> does c2w_conv_forward take packed weights there
        self._pk_want: Dict[tuple, dict] = {}   # (kind, dtype) -> {name of a matrix that is kept packed: weight version of its copy}
        self._pk_desc: Dict[tuple, torch.Tensor] = {}  # ((kind, dtype), names) -> descriptor table of one pack launch
        self._pk_sync: Dict[tuple, list] = {}  # (kind, dtype) -> [(event behind a pack launch, ids of the streams ordered behind it)]
        self.use_packed_weights = os.environ.get("C2W_NO_WPACKED") is None
        self._gwpad: Dict[str, torch.Tensor] = {}  # name -> padded fp32 weight-gradient scratch
        self._manual_ver = 0
        self._wg_stream = None  # second HIP stream for the weight-gradient launches (see _wg)
        # Input-gradient operands rebuilt on the side stream beside the next forward: part of the two-stream mode (C2W_WGRAD_STREAM=1; round
        # 4: -0.2 ms there).  On one stream -- the default -- a second active queue costs more than the 0.18 ms of passes it hides
        # (46.74-46.83 against 46.91-47.01 ms per step): off unless C2W_DG_PREFETCH=1; C2W_NO_DG_PREFETCH=1 switches it off in either mode.
        self.prefetch_dgrad = os.environ.get("C2W_NO_DG_PREFETCH") is None and \
            (os.environ.get("C2W_WGRAD_STREAM", "0") == "1" or os.environ.get("C2W_DG_PREFETCH") == "1")
        self._dg_ready = None  # event behind input-gradient operands that were rebuilt on the gradient stream (prefetch_backward_operands)
        # C2W_WGRAD_STREAM=0 (read once, here; or set the attribute): weight gradients on the caller's stream, every kernel alone on the
        # chip -- what bench.py's by_kernel pass and the serialised rocprof runs use
        # Round 5: OFF by default.  With the weight gradients of a level grouped into one launch (below) every kernel fills the chip on its
        # own, and two queues whose kernels cannot share a CU only cost each other (B = 128: 46.2 against 46.7 ms; B = 16 ... 64, 52
        # channels, fp16 and the 256x256 variant: 0.1-0.45 ms per step, profiles/r05_experiments.md section 8).  Rounds 1-4 (per-layer
        # launches) gained 2.4 ms from the second queue.  C2W_WGRAD_STREAM=1 restores it.
        self.use_grad_stream = os.environ.get("C2W_WGRAD_STREAM", "0") == "1"
        # A/B knob: gradient-stream launches are enqueued N calls late (1: behind the layer's own input gradient; 2: a
        # residual block's two weight gradients during the next block; ...)
        self._wgrad_behind = max(0, int(os.environ.get("C2W_WGRAD_BEHIND", "0") or 0))
        self._wg_pending = []
        # Weight gradients of the residual-block convs are collected per geometry during the backward and launched together at the
        # level boundaries (ops.conv_wgrad_grouped: the 6 or 12 layers of a level side share one shape).  C2W_WGRAD_GROUP=0: one
        # launch per layer (rounds 1-4); =N: only levels whose grid is at most N pixels high (default 64); =1: every level.
        # Measured at B = 128 (profiles/r05_experiments.md section 2): 64 is 0.32 ms per step ahead of 0 -- below 128x128 the per-layer
        # launches were mostly hidden beside the other queue's; grouped, they need a third of the launches and of the partial sums.
        # At 128x128 a group is neutral when it is launched at the level boundary and costs 0.9 ms when it is enqueued in front of
        # the side's last input gradient (252 workgroups that hold their CUs for 3 ms).
        self.keep_ln_stats = os.environ.get("C2W_NO_LN_STATS") != "1"  # A/B knob: fused LayerNorms hand their 1/sigma to the backward
        _grp = os.environ.get("C2W_WGRAD_GROUP", "64")
        self.group_wgrads = _grp != "0"
        self.group_wgrads_max_side = int(_grp) if _grp.isdigit() and int(_grp) > 1 else 1 << 30
        # above that side: groups of at most this many layers (1 = one launch per layer); C2W_WGRAD_GROUP_TOP (A/B)
        self.group_wgrads_top = max(1, int(os.environ.get("C2W_WGRAD_GROUP_TOP", "1")))
        self._wg_groups: Dict[tuple, list] = {}  # (geometry, dtype) -> [(x, dY, record, geometry)] not launched yet
        self._wg_group_ok: Dict[tuple, bool] = {}
        self._done_releases: list = []  # one callback per live backward_steps generator: hands on the "done" offsets it held back while a group was pending
        self._pub = None  # ops.HostRing of published scalars (publish / published)
        self._skip_dw = False  # inside backward(want_dw=False): weight-gradient launches are skipped
        self._ws: Dict[int, torch.Tensor] = {}  # stream handle -> split-K scratch of the weight-gradient launches on that stream
        self._splitk_plans: Dict[tuple, tuple] = {}  # conv geometry -> (workgroups per tile, scratch bytes) (_splitk)
        self.debug_trace: Optional[list] = None  # diagnostics: a list collects (name, output tensor[, operands of a conv]) of every conv / attention launch of a forward
        self.attach(net)

    # ------------------------------------------------------------------ parameter storage
    def attach(self, net) -> None:
        """(Re)build the flat buffer from the module's current parameters and make them views into it."""
        lay = self.layout
        params = dict(net.named_parameters())
        missing = set(lay.views) ^ set(params)
        if missing:
            raise RuntimeError(f"parameter set mismatch between module and layout: {sorted(missing)[:4]} ...")
        dev = next(iter(params.values())).device
        flat = torch.zeros(lay.numel, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for name, (off, shape, strides) in lay.views.items():
                view = torch.as_strided(flat, shape, strides, off)
                view.copy_(params[name].detach().to(torch.float32))
                # the Parameter OBJECT stays (optimizers, DDP reducers and EMA copies made before the first forward hold it, as
                # training_loop.py:116-131 does); only its storage moves into the flat buffer
                params[name].data = view
        # `p.data = view` leaves each Parameter its own version counter: writes through the Parameter objects (torch.optim steps,
        # load_state_dict, EMA copies) never bump flat._version, so the cache key folds the Parameters' counters in (_version)
        self._bound = [params[name] for name in lay.views]
        me = weakref.ref(self)
        self._bound_by_id = {id(prm): prm for prm in self._bound}
        for i in self._bound_by_id:
            _ENGINE_BY_PARAM_ID[i] = me
        self.generation = getattr(self, "generation", 0) + 1  # bumped per (re-)attach: holders of flat-layout plans re-validate
        self._slots, self._slots_net, self._links = None, (lambda: None), []
        self._offs4 = [4 * off for off, _, _ in lay.views.values()]
        self._pver = self._param_versions()
        self.flat = flat
        self.flat_grad = None
        self.shadows.clear()
        self.dg.clear()
        self.dg_lin = None
        self._shadow_ver.clear()
        self._dg_ver.clear()
        self._dg_desc.clear()
        self._wpad.clear()
        self._pk.clear()
        self._pk_tab.clear()
        self._pk_ok.clear()
        self._pk_want.clear()
        self._pk_desc.clear()
        self._pk_sync.clear()
        self._gwpad.clear()

    def is_attached(self, net) -> bool:
        """Are the module's parameters still the objects bound at attach() AND still views into the flat buffer?  Runs at the top of
        every forward (and of every EMA / optimizer call that looks the engine up), on the host's critical path when the caller
        synchronises once per step (training_loop.py:385: loss.item()): the (module, attribute) slots are resolved once per attach."""
        if self.flat is None:
            return False
        slots = self._slots
        if slots is None or self._slots_net() is not net:
            slots = self._slots = [_resolve(net, name) for name in self.layout.views]
            self._slots_net = weakref.ref(net)
            self._links = _module_links(net, self.layout.views)
        for parent, key, child in self._links:  # module surgery after the first forward (net.unet.x = new_module): the cached slots are stale
            if parent._modules.get(key) is not child:
                self._slots = None
                return False
        base = self.flat.data_ptr()
        for (mod, attr), bound, off in zip(slots, self._bound, self._offs4):
            p = mod._parameters[attr]
            if p is not bound or p.data_ptr() != base + off or p.dtype != torch.float32:
                return False
        return True

    def ensure_grad_buffer(self, net=None, bind: bool = False) -> torch.Tensor:
        """Flat fp32 gradient buffer (same offsets as the parameters).  With ``bind`` every Parameter's ``.grad`` becomes
        a view into it, so optimizers / all-reduce see one contiguous tensor."""
        if self.flat_grad is None:
            self.flat_grad = torch.zeros_like(self.flat)
        if bind and net is not None:
            for name, (off, shape, strides) in self.layout.views.items():
                mod, attr = _resolve(net, name)
                mod._parameters[attr].grad = torch.as_strided(self.flat_grad, shape, strides, off)
        return self.flat_grad

    def _param_versions(self) -> int:
        return sum(p._version for p in self._bound)

    def refresh_version(self) -> None:
        """Re-read the bound Parameters' version counters (once per forward: 228 attribute reads, not one set per launch)."""
        self._pver = self._param_versions()

    def _version(self):
        """Key of every cache derived from the weights (16-bit shadow, padded input-conv operand, input-gradient operands):
        in-place writes on the flat buffer, raw-pointer writers (weights_changed) and writes through the Parameter objects."""
        return (self.flat._version, self._manual_ver, self._pver)

    def weights_changed(self, shadow_fresh: Optional[int] = None) -> None:
        """Call after the flat buffer was rewritten through raw pointers (fused optimizer): torch's version counter
        does not see those writes.  ``shadow_fresh``: the 16-bit dtype whose shadow the writer also refreshed."""
        self._manual_ver += 1
        self.refresh_version()
        if shadow_fresh is not None and shadow_fresh in self.shadows:
            self._shadow_ver[shadow_fresh] = self._version()

    def shadow_for(self, dt: int) -> torch.Tensor:
        """The flat parameter buffer in the 16-bit format ``dt``, refreshed if the weights changed since the last cast."""
        sh = self.shadows.get(dt)
        if sh is None:
            sh = self.shadows[dt] = torch.empty(self.layout.numel, dtype=TORCH_DTYPE[dt], device=self.flat.device)
        if self._shadow_ver.get(dt) != self._version():
            ops.cast_f32(self.flat, sh, self.layout.numel, dt)
            self._shadow_ver[dt] = self._version()
        return sh

    def _w(self, rec: ConvRec, dt: int) -> torch.Tensor:
        if rec.kstride != rec.cin:  # network-input conv: the kernels want K in whole 128-byte chunks -> zero-padded operand copy
            key = (rec.name, dt)
            ent = self._wpad.get(key)
            if ent is None or ent[0] != self._version():
                buf = ent[1] if ent is not None else torch.zeros(rec.rows * rec.taps * rec.kstride, dtype=TORCH_DTYPE[dt], device=self.flat.device)
                buf.view(rec.rows, rec.taps, rec.kstride)[:, :, : rec.cin].copy_(
                    self.flat[rec.w_off: rec.w_off + rec.rows * rec.taps * rec.cin].view(rec.rows, rec.taps, rec.cin))
                self._wpad[key] = (self._version(), buf)
                ent = self._wpad[key]
            return ent[1]
        if rec.lin or dt == DTYPE_F32:
            return self.flat[rec.w_off:]
        return self.shadow_for(dt)[rec.w_off:]

    def prepare_forward(self, dt: int) -> None:
        """Build every lazily cached forward operand (16-bit weight shadow, padded input-conv weights) on the current stream,
        so that forwards issued afterwards on other streams only read them."""
        self.refresh_version()
        for rec in self.layout.convs.values():
            self._w(rec, DTYPE_F32 if rec.lin else dt)
        if self.use_packed_weights:
            # only what earlier launches asked for (the levels the 16x16-tile kernel serves at the batch sizes seen so far), one launch;
            # a matrix that a later geometry asks for first is packed on the stream that asks and the others wait on its event (_packed)
            self.repack_wanted(dt)

    def _b(self, rec: ConvRec) -> torch.Tensor:
        return self.flat[rec.b_off:]

    def _wT(self, rec: ConvRec, dt: int) -> torch.Tensor:
        """Operand of the input-gradient GEMM for ``rec`` ([cin][taps][dg_ld]); rebuilt when the weights changed."""
        lay = self.layout
        key = "lin" if rec.lin else dt
        if rec.lin:
            if self.dg_lin is None:
                self.dg_lin = torch.zeros(max(lay.dg_lin_numel, 1), dtype=torch.float32, device=self.flat.device)
            buf, d = self.dg_lin, DTYPE_F32
        else:
            if dt not in self.dg:
                self.dg[dt] = torch.zeros(max(lay.dg_numel, 1), dtype=TORCH_DTYPE[dt], device=self.flat.device)
            buf, d = self.dg[dt], dt
        if self._dg_ver.get(key, -1) != self._version():
            if key not in self._dg_desc:  # one launch for all matrices of the group: descriptor table built once
                recs = [r for r in lay.convs.values() if r.dg_off >= 0 and r.lin == rec.lin]
                tab = [v for r in recs for v in (r.w_off, r.dg_off, r.rows, r.taps, r.cin, r.cin, r.dg_ld, int(bool(r.flip)))]
                self._dg_desc[key] = (torch.tensor(tab, dtype=torch.int64, device=self.flat.device), len(recs))
            desc, n = self._dg_desc[key]
            if n:
                ops.weight_transpose_batched(self.flat, buf, desc, n, d)
            self._dg_ver[key] = self._version()
        return buf[rec.dg_off:]

    def _packed(self, kind: str, rec: ConvRec, dt: int) -> Optional[torch.Tensor]:
        """Stage-major packed copy of ``rec``'s forward (kind "f") or input-gradient (kind "d") operand in the 16-bit format ``dt``, or
        None when the matrix has none (1x1 / Linear / fp32 / the padded network-input operand).

        Addresses are STABLE: the buffer of a (kind, dtype) is laid out once for every eligible matrix (72 M elements = 144 MB of address
        space, touched only where something is packed) and never reallocated, so a captured hipGraph that reads a packed matrix
        (score_fn._score_graphed) stays valid when another launch geometry later asks for a matrix that was not packed before.  Only the
        matrices some launch asked for are kept packed (at B = 128 the levels the 16x16-tile kernel serves hold 11 of the 72 M
        parameters): one launch repacks all of them when the weights changed, one launch packs a newcomer."""
        if dt == DTYPE_F32 or rec.lin or rec.taps != 9:
            return None
        key = (kind, dt)
        tab = self._pk_tab.get(key)
        if tab is None:  # fixed offsets of every eligible matrix, in layout order
            offs, total = {}, 0
            for r in self.layout.convs.values():
                if r.lin or r.taps != 9:
                    continue
                if kind == "f":
                    if r.kstride != r.cin or r.cin % 32:
                        continue
                    ent = (r.w_off, r.rows, r.cin)
                else:
                    if r.dg_off < 0 or r.dg_ld % 32:
                        continue
                    ent = (r.dg_off, r.cin, r.dg_ld)
                offs[r.name] = (total,) + ent
                total += ops.packed_conv_weights_numel(ent[1], ent[2])
            tab = self._pk_tab[key] = (offs, total)
        offs, total = tab
        ent = offs.get(rec.name)
        if ent is None:
            return None
        if kind == "f":
            src = self.shadow_for(dt)
        else:
            self._wT(rec, dt)  # refreshes the transposed copies if the weights changed
            src = self.dg[dt]
        buf = self._pk.get(key)
        if buf is None:
            buf = self._pk[key] = torch.empty(max(total, 1), dtype=TORCH_DTYPE[dt], device=self.flat.device)
        want = self._pk_want.setdefault(key, {})  # name -> weight version its packed copy was made from
        ver = self._version()
        if want.get(rec.name) != ver:
            want[rec.name] = None
            stale = tuple(n for n, v in want.items() if v != ver)  # everything wanted after an update; the newcomer alone otherwise
            desc = self._pk_desc.get((key, stale))
            if desc is None:
                rows_t = [v for n in stale for v in (offs[n][1], offs[n][0], offs[n][2], offs[n][3])]
                desc = self._pk_desc[(key, stale)] = torch.tensor(rows_t, dtype=torch.int64, device=self.flat.device)
            ops.pack_conv_weights_batched(src, buf, desc, len(stale), dt)
            for n in stale:
                want[n] = ver
            if buf.is_cuda:  # forwards on OTHER streams (score_fn: window batches alternate between streams) must see this launch
                ev = torch.cuda.Event()
                ev.record()
                rec_ = (ev, {torch.cuda.current_stream().cuda_stream})
                # one record per pack launch that is still the newest writer of some matrix: a launch that rewrote EVERYTHING wanted
                # supersedes the older ones; a newcomer's launch joins them (a stream that packs a newcomer has not thereby waited for
                # the packs other streams made earlier)
                if len(stale) == len(want):
                    self._pk_sync[key] = [rec_]
                else:
                    self._pk_sync.setdefault(key, []).append(rec_)
        if buf.is_cuda and key in self._pk_sync:
            sid = torch.cuda.current_stream().cuda_stream
            for ev, seen in self._pk_sync[key]:
                if sid not in seen:  # once per (pack launch, stream): order this stream behind every launch that wrote copies it may read
                    torch.cuda.current_stream().wait_event(ev)
                    seen.add(sid)
        return buf[ent[0]:]

    def repack_wanted(self, dt: int) -> None:
        """Refresh, with one launch per kind, the packed copies earlier launches asked for (prepare_forward)."""
        for (kind, d), want in self._pk_want.items():
            if d == dt and want:
                self._packed(kind, self.layout.convs[next(iter(want))], dt)

    def prefetch_backward_operands(self, dt: int) -> None:
        """Rebuild the input-gradient operands (transposed copies of every weight matrix, and the packed copies the 16x16-tile launches
        asked for) NOW, on the gradient stream, behind everything enqueued so far on the current stream (= the optimizer's update):
        0.18 ms of HBM-bound passes that then run next to the following forward's matrix-core launches instead of in front of the
        backward.  ``backward`` waits for them (``_dg_ready``); without a gradient stream nothing happens and backward builds them itself."""
        side = self.side_stream() if self.prefetch_dgrad else None
        if side is None:
            return
        self.refresh_version()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for grp_lin in (False, True):
                rec0 = next((r for r in self.layout.convs.values() if r.dg_off >= 0 and r.lin == grp_lin), None)
                if rec0 is not None:
                    self._wT(rec0, DTYPE_F32 if grp_lin else dt)
            want = self._pk_want.get(("d", dt))
            if want and self.use_packed_weights:
                self._packed("d", self.layout.convs[next(iter(want))], dt)
            ev = torch.cuda.Event()
            ev.record()
        self._dg_ready = ev

    def _conv_weights(self, kind: str, rec: ConvRec, dt: int, g: dict, fused_ln_bwd: bool = False):
        """(operand, wpacked flag) for a conv launch of geometry ``g``: the packed copy where the launch goes to the 16x16-tile kernel
        (ops.conv_wpacked_supported; -4.6 % on the dominant launch, bit-identical results), else the plain one."""
        plain = self._w(rec, dt) if kind == "f" else self._wT(rec, dt)
        if not self.use_packed_weights or dt == DTYPE_F32:
            return plain, False
        key = (kind, rec.name, dt, g["B"], g["Hin"], g["Win"], g["Hout"], g["Wout"], g["Cout"], g["mode"], ops.KNOBS_GENERATION)
        ok = self._pk_ok.get(key)
        if ok is None:
            ok = self._pk_ok[key] = bool(ops.conv_wpacked_supported(g, dt))
        if not ok:
            return plain, False
        pk = self._packed(kind, rec, dt)
        return (pk, True) if pk is not None else (plain, False)

    def _gw(self, rec: ConvRec) -> torch.Tensor:
        return self.flat_grad[rec.w_off:]

    def _gb(self, rec: ConvRec) -> torch.Tensor:
        return self.flat_grad[rec.b_off:]

    # ------------------------------------------------------------------ weight gradients on a second stream
    def grad_stream(self):
        """The stream every write into ``flat_grad`` is enqueued on, or None (CPU tensors / ``use_grad_stream`` off, the default since
        round 5: the current stream).  A layer's weight gradient and its input gradient both depend only on the layer's output
        gradient; with one launch per layer (rounds 1-4) the chip idled through every kernel's last round of workgroups, the split-K
        reduction launches and the launch gaps on one stream, and a second stream filled those holes (the LDS footprints forbid real
        co-residency: 160 KB + 2 x 71.7 KB > 160 KB).  Grouped launches leave no such holes."""
        if self.flat is None or not self.flat.is_cuda or not self.use_grad_stream:
            return None
        return self.side_stream()

    def side_stream(self):
        """A second HIP stream on another hardware queue than the caller's (streams.py), or None on the CPU: the gradient stream when
        ``use_grad_stream`` is on, and in any case where the next backward's operands are rebuilt beside the forward
        (prefetch_backward_operands)."""
        if self.flat is None or not self.flat.is_cuda:
            return None
        if self._wg_stream is None or self._wg_stream.device != self.flat.device:
            from .streams import independent_stream
            # on another hardware queue than the caller's stream (streams.py); C2W_GRAD_STREAM_PRIORITY: A/B knob (HIP: -1 high, 0 normal,
            # 1 low where the runtime offers it)
            self._wg_stream = independent_stream(self.flat.device, priority=int(os.environ.get("C2W_GRAD_STREAM_PRIORITY", "0")))
        return self._wg_stream

    def publish(self, scalar: torch.Tensor):
        """Enqueue, behind the launch that produced the fp32 device scalar ``scalar``, its copy into a slot of pinned host memory
        (ops.HostRing); returns (slot, sequence number) for ``published``."""
        if self._pub is None:
            self._pub = ops.HostRing()
        return self._pub.publish(scalar)

    def published(self, slot: int, seq: int, timeout_s: float = 20.0):
        """The value of publication ``seq`` as a Python float once the device has written it (polling host memory: no stream is
        synchronised), or None if the slot has been reused or nothing arrived in ``timeout_s``."""
        import struct
        bits = self._pub.read_bits(slot, seq, timeout_s)
        return None if bits is None else struct.unpack("<f", struct.pack("<i", bits))[0]

    def _on_grad_stream(self, fn, *tensors) -> None:
        """Run ``fn`` (launches that read ``tensors`` and write gradient memory) on the gradient stream, behind everything
        enqueued so far on the current stream.  The tensors were allocated on the current stream and may be dropped by the
        caller before the gradient stream has read them: record_stream defers the reuse of their memory."""
        side = self.grad_stream()
        if side is None:
            fn()
            return
        if self._wgrad_behind:  # experiment knob: this launch is enqueued N calls later, i.e. behind the input gradients issued in between
            self._wg_pending.append((fn, tensors))
            while len(self._wg_pending) > self._wgrad_behind:
                self._issue_on(side, *self._wg_pending.pop(0))
            return
        self._issue_on(side, fn, tensors)

    @staticmethod
    def _issue_on(side, fn, tensors) -> None:
        side.wait_stream(torch.cuda.current_stream())
        for t in tensors:
            t.record_stream(side)
        with torch.cuda.stream(side):
            fn()

    def join_grad_stream(self) -> None:
        """Make the current stream wait for every gradient launch enqueued so far."""
        self.flush_wgrad_groups()
        side = self.grad_stream()
        if side is not None:
            while self._wg_pending:
                self._issue_on(side, *self._wg_pending.pop(0))
            torch.cuda.current_stream().wait_stream(side)

    def workspace(self, min_bytes: int = 0) -> Optional[torch.Tensor]:
        """Split-K scratch for a weight-gradient launch on torch's CURRENT stream.  One buffer per (engine, stream): launches on one
        stream are ordered, so they can share it; two engines, or one engine's backward on two streams (training on the gradient
        stream next to an exact-guidance backward elsewhere), never see each other's partial sums.  ``min_bytes``: a grouped launch's
        need (ops.conv_wgrad_grouped_workspace_bytes); a larger buffer replaces the stream's (the old one is freed behind the launches
        that used it: same stream)."""
        if self.flat is None or not self.flat.is_cuda:
            return None
        key = torch.cuda.current_stream(self.flat.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.device != self.flat.device or ws.numel() * 4 < min_bytes:
            ws = self._ws[key] = ops.new_workspace(self.flat.device, max(ops.WORKSPACE_BYTES, (min_bytes + (1 << 20) - 1) >> 20 << 20))
        return ws

    def _splitk(self, g: dict, dt: int, act: int):
        """(scratch, workgroups per tile) if this inference launch should deal its K chunks to several workgroups (ops.conv_splitk_plan:
        fewer output tiles than the chip has CUs -- the deep levels of a sampler step on a short trajectory), else None.  The plan is
        a pure function of geometry and knobs: remembered per geometry."""
        if not self.use_splitk or self.flat is None or not self.flat.is_cuda:
            return None
        key = (g["B"], g["Hin"], g["Win"], g["Cin"], g["Hout"], g["Wout"], g["Cout"], g["ldy"], g["wrows"], g["mode"], dt, act, ops.KNOBS_GENERATION)
        plan = self._splitk_plans.get(key)
        if plan is None:
            plan = self._splitk_plans[key] = ops.conv_splitk_plan(g, dt, act)
        ns, nbytes = plan
        if ns <= 1:
            return None
        return self.workspace(nbytes), ns

    def _wg(self, x: torch.Tensor, gy: torch.Tensor, rec: ConvRec, g: dict, dt: int, group: bool = False) -> None:
        """dW, dbias of ``rec`` (dense operand) into the flat gradient buffer, on the gradient stream.  ``group``: the layer is one of
        several with this geometry whose output gradients appear one after the other (the residual-block convs of a level side): it is
        queued and launched with the others at the next flush_wgrad_groups()."""
        if self._skip_dw:
            return
        # "up to N x N grids" is meant at the reference's batch of 128 per GPU: what decides is the layer's K extent (B x H x W pixels;
        # the deep variant's 128x128 level at B = 32 is the default network's 64x64 level at B = 128)
        npx = g["B"] * g["Hout"] * g["Wout"]
        small = npx <= 128 * self.group_wgrads_max_side ** 2
        cap = 16 if small else self.group_wgrads_top
        if not small and cap == 1 and npx <= 256 * self.group_wgrads_max_side ** 2:
            # one size up -- the 128x128 level at 64 windows per GPU, the reference's global batch of 512 on 8 GPUs: groups of three
            # (round 6, B = 64: 26.00 -> 25.84 ms per step; at B = 128 the same grouping costs 0.25 ms: profiles/r05_experiments.md)
            cap = 3
        if group and self.group_wgrads and cap > 1:
            key = (g["B"], g["Hin"], g["Win"], g["Cin"], g["Hout"], g["Wout"], g["Cout"], g["ldy"], g["mode"], dt, ops.KNOBS_GENERATION)
            ok = self._wg_group_ok.get(key)
            if ok is None:
                ok = self._wg_group_ok[key] = bool(ops.conv_wgrad_grouped_supported(g, 2, dt))
            if ok:
                lst = self._wg_groups.setdefault(key, [])
                lst.append((x, gy, rec, g))
                if len(lst) >= cap:
                    self._flush_group(key)
                    if not self._wg_groups:
                        self._release_done()
                return
        self._on_grad_stream(lambda: ops.conv_wgrad(x, gy, self._gw(rec), g, dt, dbias=self._gb(rec), workspace=self.workspace()), x, gy)

    def _flush_group(self, key: tuple) -> None:
        lst = self._wg_groups.pop(key, None)
        if not lst:
            return
        g, dt = lst[0][3], key[9]
        if len(lst) == 1 or not ops.conv_wgrad_grouped_supported(g, len(lst), dt):
            for x, gy, rec, gi in lst:
                self._on_grad_stream(lambda x=x, gy=gy, rec=rec, gi=gi: ops.conv_wgrad(x, gy, self._gw(rec), gi, dt, dbias=self._gb(rec),
                                                                                    workspace=self.workspace()), x, gy)
            return

        def run():
            ws = self.workspace(ops.conv_wgrad_grouped_workspace_bytes(g, len(lst), dt))
            ops.conv_wgrad_grouped([(x, gy, self._gw(rec), self._gb(rec)) for x, gy, rec, _ in lst], g, dt, workspace=ws)
        self._on_grad_stream(run, *[t for x, gy, _, _ in lst for t in (x, gy)])

    def flush_wgrad_groups(self) -> None:
        """Launch every queued weight gradient (one launch per geometry) and hand on the "done" offsets that were held back for them."""
        for key in list(self._wg_groups):
            self._flush_group(key)
        self._release_done()

    def _release_done(self) -> None:
        for rel in list(self._done_releases):
            rel()

    def _wgrad(self, rec: ConvRec, x: torch.Tensor, gy: torch.Tensor, g: dict, dt: int) -> None:
        """dW (+ dbias) of ``rec`` into the flat gradient buffer; a padded-operand layer goes through a padded scratch."""
        if self._skip_dw:
            return
        if rec.kstride == rec.cin:
            self._wg(x, gy, rec, g, dt)
            return

        def run():
            n = rec.rows * rec.taps * rec.kstride
            buf = self._gwpad.get(rec.name)
            if buf is None or buf.device != self.flat.device:
                buf = self._gwpad[rec.name] = torch.zeros(n, dtype=torch.float32, device=self.flat.device)
            else:
                buf.zero_()
            ops.conv_wgrad(x, gy, buf, g, dt, dbias=self._gb(rec), workspace=self.workspace())
            self.flat_grad[rec.w_off: rec.w_off + rec.rows * rec.taps * rec.cin].view(rec.rows, rec.taps, rec.cin).add_(
                buf.view(rec.rows, rec.taps, rec.kstride)[:, :, : rec.cin])
        self._on_grad_stream(run, x, gy)

    # ------------------------------------------------------------------ small helpers
    @staticmethod
    def _geom(B, Hin, Win, Cin, Hout, Wout, Cout, ldy, wrows, mode):
        return dict(B=B, Hin=Hin, Win=Win, Cin=Cin, Hout=Hout, Wout=Wout, Cout=Cout, ldy=ldy, wrows=wrows, mode=mode)

    def _linear(self, name: str, x: torch.Tensor, rows: int, act: int, tape: Optional[Tape], need_dx: bool = True) -> torch.Tensor:
        """fp32 GEMM  y[rows_x][out] = x . W^T + b  on the fp32 matrix-core path (Linear layers of model/score.py:56-57,
        model/nn.py:149)."""
        rec = self.layout.convs[name]
        y = torch.empty((rows, rec.rows), dtype=torch.float32, device=x.device)
        if rows == 1 and tape is None and self.use_gemv and act in (ACT_NONE, ACT_SILU, ACT_RELU):
            # one t for the whole batch (the sampler): a matrix-vector product instead of a 16-pixel MFMA tile with one live column
            ops.gemv_f32(x, self._w(rec, DTYPE_F32), self._b(rec), y, rec.rows, rec.cin, rec.kstride, act)
            return y
        g = self._geom(rows, 1, 1, rec.kstride, 1, 1, rec.rows, rec.rows, rec.rows, CONV_1X1)
        ops.conv(x, self._w(rec, DTYPE_F32), self._b(rec), y, g, DTYPE_F32, act=act)
        if tape is not None:
            def bw(gy: torch.Tensor) -> Optional[torch.Tensor]:
                self._wgrad(rec, x, gy, g, DTYPE_F32)  # (a padded operand -- map_forcing -- goes through its padded scratch)
                if not need_dx:
                    tape.done(rec.w_off)
                    return None
                if rec.rows >= self.LINEAR_DGRAD_SPLIT_MIN_ROWS and rec.kstride == rec.cin:
                    # few rows, very long reduction (proj: B x 8448 -> 512): as a forward GEMM that is 4 workgroups walking
                    # K = 8448.  It is also the weight-gradient GEMM of a 1x1 conv over rec.rows "pixels" with the weight
                    # matrix as the input and gy^T as the output gradient -- which splits the reduction over the whole chip.
                    ld = (rows + 3) // 4 * 4
                    gyT = torch.zeros((rec.rows, ld), dtype=torch.float32, device=x.device)
                    # LDS-tiled transpose (torch's strided copy of the 128 x 8448 matrix took 0.49 ms per step)
                    ops.weight_transpose(gy, gyT, rows, 1, rec.rows, rec.rows, ld, False, DTYPE_F32)
                    dx = torch.zeros((rows, rec.cin), dtype=torch.float32, device=x.device)
                    gt = self._geom(rec.rows, 1, 1, rec.cin, 1, 1, rows, ld, rows, CONV_1X1)
                    # same kernels, same split-K workspace as the weight gradients: same stream, then wait for the result
                    wmat = self._w(rec, DTYPE_F32)
                    self._on_grad_stream(lambda: ops.conv_wgrad(wmat, gyT, dx, gt, DTYPE_F32, workspace=self.workspace()), gyT, dx)
                    self.join_grad_stream()
                    # only now: this route reads the layer's weights from the flat buffer itself, and "done" may start the
                    # optimizer on them (Trainer: the update chases the backward)
                    tape.done(rec.w_off)
                    return dx
                dx = torch.empty((rows, rec.cin), dtype=torch.float32, device=x.device)
                gd = self._geom(rows, 1, 1, rec.dg_ld, 1, 1, rec.cin, rec.cin, rec.cin, CONV_1X1)
                ops.conv(gy, self._wT(rec, DTYPE_F32), None, dx, gd, DTYPE_F32)
                tape.done(rec.w_off)
                return dx
            tape.steps.append(bw)
        return y

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, t: torch.Tensor, dt: int, tape: Optional[Tape] = None, noise: Optional[Tuple] = None,
                want_dx: bool = False, nhwc_out: bool = False, x_nhwc: Optional[torch.Tensor] = None, shape=None,
                forcing: Optional[torch.Tensor] = None, fold: Optional[dict] = None, loss: Optional[dict] = None):
        """eps_pred = ScoreUNet(x, t).  x: (B,C,H,W) fp32 on the GPU; t: numel 1 or B.
        ``fold`` (inference, with ``nhwc_out``): dict(segs=[(eps trajectory (L,F,H,W) fp32, first window, count, first batch row)], k, F,
        nwin) -- the batch rows are windows of trajectories and only what src/thor/score.py:76-88 keeps of them is wanted: where the
        kernel exists the output convolution computes the centre frame's F rows only and writes them into the trajectories
        (ops.conv_center; the other frames of a trajectory's first / last window through the full convolution of that one window) and
        None is returned; otherwise the NHWC output rows are returned and the caller scatters them.
        noise = (eps, musig): fuse the forward noise process x_t = mu x + sigma eps into the input conversion; eps may be an int
        seed instead of a tensor: the kernel regenerates the Philox stream of that seed (ops.philox_normal) and eps never exists.
        ``loss`` (training, with ``nhwc_out`` and a noise SEED): dict(sum, gscale[, scaler]) -- fuse the loss tail of src/thor/pipelines.py:35
        into the output convolution where the kernel exists (ops.conv_loss_supported): the input conversion keeps the noise it mixes in
        (half-precision rows), the returned rows are dY = (prediction - eps) * gscale, ``sum`` has received sum (prediction - eps)^2 and
        ``tape.meta["loss_fused"]`` is True; otherwise nothing changes.
        With ``tape`` every op records its backward closure (training / exact guidance)."""
        lay = self.layout
        self.refresh_version()
        T = TORCH_DTYPE[dt]
        if x_nhwc is not None:  # rows already in the network's input layout (the sampler's fused window gather)
            B, C, H, W = shape
            dev = x_nhwc.device
        else:
            B, C, H, W = x.shape
            dev = x.device
        L = len(lay.levels)
        if C != lay.in_channels:
            raise ValueError(f"expected {lay.in_channels} channels, got {C}")
        if H % (1 << (L - 1)) or W % (1 << (L - 1)):
            raise ValueError("spatial size must be divisible by 2**(levels-1) (skip connections, model/nn.py:238)")
        ck = ops.CK[dt]
        for lv in lay.levels:
            if lv.channels % ck:
                raise ValueError(f"hidden_channels must be multiples of {ck} for this compute dtype")
        train = tape is not None
        lazy = x_nhwc is None and hasattr(x, "materialize")  # data.WindowBatch: windows still inside the dataset array
        if x_nhwc is None and not lazy:
            x = x.contiguous().float()
        tt = t.reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
        Bt = tt.numel()
        if Bt not in (1, B):
            raise ValueError("t must hold 1 or B values (model/score.py:60)")
        ldm = lay.sum_c if (Bt == B and B > 1) else 0

        # ---- time embedding MLP + all modulation vectors (fp32)
        pe = torch.empty((Bt, lay.noise_features), dtype=torch.float32, device=dev)
        ops.timestep_embedding(tt, pe, Bt, lay.noise_features)
        emb = self._mlp_layer("map_layer0", pe, Bt, tape, need_dx=False)
        zf = None
        if lay.forcing_dim:  # emb = silu(map_layer1(.) + map_forcing(forcing))  (model/score.py:64-67)
            if forcing is None:
                raise ValueError("forcing_dim > 0: the forcing vector is required")
            rf = lay.convs["map_forcing"]
            fr = forcing.reshape(-1, lay.forcing_dim).to(device=dev, dtype=torch.float32)
            if fr.shape[0] not in (1, Bt) and not (Bt == 1 and fr.shape[0] == B):
                raise ValueError(f"forcing has {fr.shape[0]} rows for {Bt} time values / {B} batch items")
            if Bt == 1 and fr.shape[0] == B and B > 1:  # scalar t, per-item forcing: the embedding becomes per item
                raise NotImplementedError("per-item forcing with a scalar t: pass t with one value per batch item")
            fpad = torch.zeros((Bt, rf.kstride), dtype=torch.float32, device=dev)
            fpad[:, : lay.forcing_dim] = fr if fr.shape[0] == Bt else fr.expand(Bt, -1)
            zf = self._linear("map_forcing", fpad, Bt, ACT_NONE, None)  # not on the tape: its gradient is map_layer1's pre-activation gradient
            g_f = self._geom(Bt, 1, 1, rf.kstride, 1, 1, rf.rows, rf.rows, rf.rows, CONV_1X1)

            def forcing_bw(gz, rf=rf, fpad=fpad, g_f=g_f):
                self._wgrad(rf, fpad, gz, g_f, DTYPE_F32)
                tape.done(rf.w_off)
        elif forcing is not None:
            raise ValueError("forcing passed to a network built with forcing_dim == 0 (model/score.py:60)")
        emb = self._mlp_layer("map_layer1", emb, Bt, tape, add=zf, add_bw=forcing_bw if zf is not None and tape is not None else None)
        m_all = self._linear("proj", emb, Bt, ACT_NONE, tape)
        if train:
            dm_all = torch.zeros_like(m_all)
            tape.meta["dm_all"] = dm_all
        else:
            dm_all = None

        # ---- network input -> NHWC
        loss_rows = None  # (noise rows, channel stride) the input conversion kept for the fused loss tail
        if x_nhwc is not None:
            x0 = x_nhwc
        else:
            x0 = torch.empty((B * H * W, lay.cin_pad), dtype=T, device=dev)
            regen = noise is not None and isinstance(noise[0], int)  # regenerated noise: (seed, musig)
            done = False
            # Fused loss tail (round 6): where the output conv takes it (ops.conv_loss_supported), the input conversion KEEPS the noise it
            # mixes in -- rounded to half precision, as NHWC rows -- and the output conv's epilogue reads it back: the generator runs
            # once per step instead of twice and the prediction is never written.  The step's noise is then the rounded stream.
            if loss is not None and train and nhwc_out and regen and self.fuse_loss and dt != DTYPE_F32 and lay.out_channels == C and (H * W) % 4 == 0:
                rec_o = lay.convs["unet." + lay.levels[0].tail_key]
                g_o = self._geom(B, H, W, rec_o.kstride, H, W, lay.cout_pad, lay.cout_pad, rec_o.rows, CONV_S1)
                if ops.conv_loss_supported(g_o, dt):
                    lde = _round_up(C, 8)
                    erows = torch.empty((B * H * W, lde), dtype=torch.float16, device=dev)
                    src, offs = (x.data, x.offsets()) if lazy else (x, None)
                    if (not lazy or (x.data.is_contiguous() and x.data.dtype == torch.float32)) and \
                            ops.nchw_to_nhwc_noise_rows(src, offs, noise[0], noise[1], x0, erows, B, C, H * W, lay.cin_pad, lde, dt):
                        done = True
                        loss_rows = (erows, lde)
            if done:
                pass
            elif lazy:  # windows still inside the dataset array: convert them in place where the fused kernel takes the shape
                done = regen and x.data.is_contiguous() and x.data.dtype == torch.float32 and \
                    ops.windows_to_nhwc_noise(x.data, x.offsets(), noise[0], noise[1], x0, B, C, H * W, lay.cin_pad, dt)
                if not done:
                    x = x.materialize().contiguous().float()
            if done:
                pass
            elif regen:
                if not ops.nchw_to_nhwc_noise(x, noise[0], noise[1], x0, B, C, H * W, lay.cin_pad, dt):
                    eps = torch.empty_like(x)
                    ops.philox_normal(eps, eps.numel(), noise[0])
                    ops.nchw_to_nhwc(x, eps, noise[1], x0, B, C, H * W, lay.cin_pad, dt)
            else:
                ops.nchw_to_nhwc(x, noise[0] if noise else None, noise[1] if noise else None, x0, B, C, H * W, lay.cin_pad, dt)

        def conv3(name, xin, Hi, Wi, Ho, Wo, mode, act=ACT_NONE, res=None, ldy=None, cout=None, y2=None, want_ln=None, loss=None,
                  resn=None, no_y=False):
            """want_ln: None, or the consumer's LayerNorm to emit from this conv's epilogue: ("mod", modulation rows) for a
            residual block, ("plain", None) for an up-block.  Returns (y, geometry, record[, LN output or None]).
            The chain form (res_block): ``resn`` = dict(rstd, mean, m) -- ``res`` holds normalised rows and the residual is rebuilt from
            them; ``no_y`` -- the result is not written (y is returned as None), the emitted LayerNorm keeps its mean next to its 1/sigma."""
            rec = lay.convs[name]
            ldy_ = ldy or rec.rows
            y = torch.empty((B * Ho * Wo, ldy_), dtype=T, device=dev) if not no_y else None
            g = self._geom(B, Hi, Wi, rec.kstride, Ho, Wo, cout or rec.rows, ldy_, rec.rows, mode)
            hn = None
            lnf = None
            if want_ln is not None and act == ACT_NONE and y2 is None and ops.conv_lnfwd_supported(g, dt):
                hn = torch.empty((B * Ho * Wo, ldy_), dtype=T, device=dev)
                lnf = dict(y=hn, m=want_ln[1], ldm=ldm if want_ln[1] is not None else 0, eps=LN_EPS, unbiased=self.ln_unbiased)
                if train and self.keep_ln_stats and want_ln[0] == "mod":
                    # training: the epilogue also leaves every pixel row's 1/sigma; the block's backward then takes its LayerNorm
                    # statistics from here and the normalised rows (kept anyway: conv1's input) instead of recomputing both (res_block)
                    lnf["rstd"] = hn._c2w_rstd = torch.empty((B * Ho * Wo,), dtype=torch.float32, device=dev)
                if no_y:
                    lnf["mean"] = hn._c2w_mean = torch.empty((B * Ho * Wo,), dtype=torch.float32, device=dev)
            if no_y or resn is not None:  # (the rebuilt residual lives in the LayerNorm-emitting epilogue; an output that is not written needs its statistics kept)
                assert lnf is not None and (not no_y or "rstd" in lnf), "chain form without a fused LayerNorm (run_blocks decides both from the same answers)"
            # padded operand (network input at C = 65: rows of 128 channels): channels >= rec.cin are zero in x and in w -- a promise the
            # 16x16-tile kernel turns into fewer K steps
            wop, wpk = self._conv_weights("f", rec, dt, g)
            ops.conv(xin, wop, self._b(rec), y if y is not None else hn, g, dt, act=act, res=res, y2=y2, lnf=lnf,
                     kvalid=rec.cin if rec.kstride != rec.cin else 0, wpacked=wpk, loss=loss, resn=resn, no_y=no_y,
                     splitk=self._splitk(g, dt, act) if (not train and lnf is None and y2 is None and loss is None and not wpk) else None)
            if self.debug_trace is not None and y is not None:
                self.debug_trace.append((name, y, dict(x=xin, w=self._w(rec, dt), g=g, act=act, res=res)))
                if hn is not None:
                    self.debug_trace.append((name + " [LayerNorm emitted]", hn))
            if want_ln is not None:
                return y, g, rec, hn
            return y, g, rec

        def dgrad(rec, gy, Hi, Wi, Ho, Wo, mode, ld_out, mul=None, res=None, ln=None, mulmode=MUL_DSILU):
            """input gradient = implicit GEMM over gy with the transposed (and flipped) weights; (Hi,Wi) = gy's grid.
            ``ln``: LayerNorm-backward arguments to fuse into the epilogue; returns None when the kernel cannot fuse them."""
            g = self._geom(B, Hi, Wi, rec.dg_ld, Ho, Wo, ld_out, ld_out, rec.cin, mode)
            if ln is not None and not ops.conv_lnbwd_supported(g, dt):
                return None
            dx = torch.empty((B * Ho * Wo, ld_out), dtype=T, device=dev)
            # output conv at C = 65: gy rows are padded to dg_ld = 128 channels, the padding is zero (mse_loss_grad) and so are the
            # operand's columns there
            wop, wpk = self._conv_weights("d", rec, dt, g, fused_ln_bwd=ln is not None)
            ops.conv(gy, wop, None, dx, g, dt, res=res, mul=mul, mulmode=mulmode, ln=ln,
                     kvalid=rec.rows if rec.dg_ld != rec.rows else 0, wpacked=wpk)
            return dx

        def res_block(b: BlockSpec, xin, Hc, Wc, h0=None, want_ln=None, elide=False):
            """h0: LN(xin + m) if the producer of xin already emitted it; want_ln: the consumer's LayerNorm to emit from
            conv2's epilogue (see conv3).  Returns (block output, consumer's LN input or None).
            The chain form (round 6; training, 16-bit, 128-channel levels on the 16x16-tile kernel): ``xin`` None -- the previous block did
            not write its output; this block's residual is rebuilt inside conv2's epilogue from h0 and the statistics that came with it
            (x = h0 / rstd + mean - m); ``elide`` -- this block does not write ITS output either (returned as None): the next block of the
            side is its only reader besides the LayerNorm emitted here."""
            p = "unet." + b.key
            Cc = b.channels
            npix = B * Hc * Wc
            m = m_all.view(-1)[b.mod_offset:]
            rstd0 = getattr(h0, "_c2w_rstd", None) if h0 is not None else None
            if xin is None:
                assert rstd0 is not None and getattr(h0, "_c2w_mean", None) is not None
            if h0 is None:
                h0 = torch.empty((npix, Cc), dtype=T, device=dev)
                ops.ln_forward(xin, m, h0, npix, Hc * Wc, Cc, ldm, LN_EPS, self.ln_unbiased, dt)
            # training: conv1's epilogue writes silu(a) for the next conv and silu'(a) for the backward pass; the pre-activation
            # itself is never stored
            d1 = torch.empty((npix, Cc), dtype=T, device=dev) if train else None
            mulmode = MUL_PLAIN
            act_inf, act_train = (ACT_RELU, ACT_RELU_PAIR) if lay.activation == "relu" else (ACT_SILU, ACT_SILU_PAIR)
            h1, g1, r1 = conv3(p + ".residue.1", h0, Hc, Wc, Hc, Wc, CONV_S1, act=act_train if train else act_inf, y2=d1)
            if want_ln is not None:
                resn = dict(rstd=rstd0, mean=h0._c2w_mean, m=m) if xin is None else None
                out, g2, r2, hn = conv3(p + ".residue.3", h1, Hc, Wc, Hc, Wc, CONV_S1, res=xin if xin is not None else h0, want_ln=want_ln,
                                        resn=resn, no_y=elide)
            else:
                (out, g2, r2), hn = conv3(p + ".residue.3", h1, Hc, Wc, Hc, Wc, CONV_S1, res=xin), None
            if train:
                def bw(gy):
                    self._wg(h1, gy, r2, g2, dt, group=True)
                    da1 = dgrad(r2, gy, Hc, Wc, Hc, Wc, CONV_S1, Cc, mul=d1, mulmode=mulmode)
                    self._wg(h0, da1, r1, g1, dt, group=True)
                    dm = dm_all.view(-1)[b.mod_offset:]
                    # conv1's input gradient feeds LN's backward directly: fused into the conv epilogue where the kernel
                    # holds whole channel rows (128-channel levels in bf16), a separate pass otherwise
                    if rstd0 is not None:  # the producer's epilogue kept the statistics: h0 = the normalised rows, rstd0 their 1/sigma
                        lnb = dict(x=h0, rstd=rstd0, m=None, dm=dm, ldm=ldm, eps=LN_EPS, unbiased=self.ln_unbiased)
                    else:
                        lnb = dict(x=xin, m=m, dm=dm, ldm=ldm, eps=LN_EPS, unbiased=self.ln_unbiased)
                    dx = dgrad(r1, da1, Hc, Wc, Hc, Wc, CONV_S1, Cc, res=gy, ln=lnb)
                    if dx is None:
                        assert xin is not None, "chain form: the fused LayerNorm backward this block was built on is gone (knobs changed between forward and backward?)"
                        dh0 = dgrad(r1, da1, Hc, Wc, Hc, Wc, CONV_S1, Cc)
                        dx = torch.empty_like(dh0)
                        ops.ln_backward(dh0, xin, m, gy, dx, dm, npix, Hc * Wc, Cc, ldm, LN_EPS, self.ln_unbiased, dt)
                    tape.done(r1.w_off)  # behind the block's last launch: "done" = gradients final AND weights (copies included) no longer read
                    return dx
                tape.steps.append(bw)
            return out, hn

        def attn_block(b: BlockSpec, xin, Hc, Wc):
            p = "unet." + b.key
            Cc = b.channels
            Tn = Hc * Wc
            npix = B * Tn
            rq, rp = lay.convs[p + ".qkv"], lay.convs[p + ".proj_out"]
            hl = torch.empty((npix, Cc), dtype=T, device=dev)
            ops.ln_forward(xin, None, hl, npix, Tn, Cc, 0, LN_EPS, self.ln_unbiased, dt)
            qkv = torch.empty((npix, 3 * Cc), dtype=T, device=dev)
            gq = self._geom(npix, 1, 1, Cc, 1, 1, 3 * Cc, 3 * Cc, 3 * Cc, CONV_1X1)
            ops.conv(hl, self._w(rq, dt), self._b(rq), qkv, gq, dt)
            o = torch.empty((npix, Cc), dtype=T, device=dev)
            lse = torch.empty((npix,), dtype=torch.float32, device=dev) if train else None
            ops.attention_forward(qkv, o, lse, B, Tn, Cc, dt)
            if self.debug_trace is not None:
                self.debug_trace += [(p + " qkv", qkv), (p + " attention", o)]
            out = torch.empty((npix, Cc), dtype=T, device=dev)
            gp = self._geom(npix, 1, 1, Cc, 1, 1, Cc, Cc, Cc, CONV_1X1)
            ops.conv(o, self._w(rp, dt), self._b(rp), out, gp, dt, res=xin)
            if train:
                def bw(gy):
                    self._wg(o, gy, rp, gp, dt, group=True)  # the six proj_out / six qkv weight gradients of the level: one launch each
                    do = torch.empty((npix, Cc), dtype=T, device=dev)
                    ops.conv(gy, self._wT(rp, dt), None, do, self._geom(npix, 1, 1, Cc, 1, 1, Cc, Cc, Cc, CONV_1X1), dt)
                    dqkv = torch.empty_like(qkv)
                    delta = torch.empty((npix,), dtype=torch.float32, device=dev)
                    ops.attention_backward(qkv, o, do, lse, delta, dqkv, B, Tn, Cc, dt)
                    self._wg(hl, dqkv, rq, gq, dt, group=True)
                    dhl = torch.empty((npix, Cc), dtype=T, device=dev)
                    ops.conv(dqkv, self._wT(rq, dt), None, dhl, self._geom(npix, 1, 1, 3 * Cc, 1, 1, Cc, Cc, Cc, CONV_1X1), dt)
                    tape.done(rq.w_off)
                    dx = torch.empty_like(dhl)
                    ops.ln_backward(dhl, xin, None, gy, dx, None, npix, Tn, Cc, 0, LN_EPS, self.ln_unbiased, dt)
                    return dx
                tape.steps.append(bw)
            return out

        # ---- descent
        Hc, Wc = H, W
        lv0 = lay.levels[0]
        def mod_of(b: BlockSpec):
            return ("mod", m_all.view(-1)[b.mod_offset:])

        # the network-input conv emits the first residual block's LayerNorm input from its epilogue, like every block's second conv
        first0 = lv0.descent[0] if lv0.descent else None
        if first0 is not None and first0.kind == "res":
            cur, g_h0, r_h0, hn0 = conv3("unet." + lv0.head_key, x0, H, W, H, W, CONV_S1, want_ln=mod_of(first0))
        else:  # nothing consumes a LayerNorm of the head conv's output: do not ask the kernel for one
            (cur, g_h0, r_h0), hn0 = conv3("unet." + lv0.head_key, x0, H, W, H, W, CONV_S1), None
        if train:
            def bw_head0(gy, x0=x0, g=g_h0, rec=r_h0):
                self.flush_wgrad_groups()
                self._wgrad(rec, x0, gy, g, dt)
                dx0 = dgrad(rec, gy, H, W, H, W, CONV_S1, lay.cin_pad) if want_dx else None
                tape.done(rec.w_off)
                return dx0
            tape.steps.append(bw_head0)
        def chain_ok(Cc, Hc, Wc):
            """Do conv2 (LayerNorm emission with lnf_mean / rebuilt residual / no output) and the next block's fused LayerNorm backward
            exist for a residual block of this level?  One answer per level geometry."""
            g = self._geom(B, Hc, Wc, Cc, Hc, Wc, Cc, Cc, Cc, CONV_S1)
            return dt != DTYPE_F32 and ops.conv_lnfwd_chain_supported(g, dt) and ops.conv_lnbwd_supported(g, dt)

        def run_blocks(blocks, cur, Hc, Wc, h0, tail_ln):
            """The blocks of one level side in order.  Each residual block asks its producer -- the previous block's second
            conv -- for its LayerNorm input; ``tail_ln`` is what the consumer after the last block wants.  Returns the
            output and that consumer's LN input (None if it was not fused)."""
            hn = h0

            def want_of(j):  # the LayerNorm block j's second conv emits for its consumer
                nb = blocks[j + 1] if j + 1 < len(blocks) else None
                return (mod_of(nb) if nb.kind == "res" else None) if nb is not None else tail_ln
            for j, b in enumerate(blocks):
                if b.kind == "res":
                    nb = blocks[j + 1] if j + 1 < len(blocks) else None
                    want = want_of(j)
                    # chain form: this block's output has no reader but the next block of the side (its LayerNorm comes out of this
                    # block's conv2, its residual add can rebuild the sum) -- where the kernels exist, it is not written.  The rebuilding
                    # lives in the LayerNorm-emitting epilogue, so the NEXT block's conv2 must emit one too (a side's last block does
                    # only in front of an up-block)
                    elide = train and self.chain_blocks and self.keep_ln_stats and nb is not None and nb.kind == "res" and \
                        want_of(j + 1) is not None and chain_ok(b.channels, Hc, Wc)
                    cur, hn = res_block(b, cur, Hc, Wc, h0=hn, want_ln=want, elide=elide)
                else:
                    cur, hn = attn_block(b, cur, Hc, Wc), None
            return cur, hn

        skips: List[torch.Tensor] = []
        for i, lv in enumerate(lay.levels):
            if i > 0:
                xin = cur
                Hp, Wp = Hc, Wc
                Hc, Wc = Hc // 2, Wc // 2
                cur, g_h, r_h = conv3("unet." + lv.head_key, xin, Hp, Wp, Hc, Wc, CONV_S2)
                if train:
                    def bw_head(gy, xin=xin, g=g_h, rec=r_h, Hp=Hp, Wp=Wp, Hc=Hc, Wc=Wc, lvl=i - 1):
                        self.flush_wgrad_groups()  # the level below is complete: its residual-block weight gradients go out together
                        self._wg(xin, gy, rec, g, dt)
                        # dx of the stride-2 conv + the gradient that arrived through the skip connection (model/nn.py:238)
                        dxs = dgrad(rec, gy, Hc, Wc, Hp, Wp, CONV_TS2, rec.cin, res=tape.gskip.pop(lvl))
                        tape.done(rec.w_off)
                        return dxs
                    tape.steps.append(bw_head)
            cur, _ = run_blocks(lv.descent, cur, Hc, Wc, hn0 if i == 0 else None, None)
            if i < L - 1:
                skips.append(cur)
        # ---- ascent
        h0_carry = None  # LN input of the level's first block when the up-conv below already produced it
        for i in reversed(range(L)):
            lv = lay.levels[i]
            cur, hl_ready = run_blocks(lv.ascent, cur, Hc, Wc, h0_carry, ("plain", None) if i > 0 else None)
            h0_carry = None
            if i > 0:
                xin = cur
                Cc = lv.channels
                npix = B * Hc * Wc
                if hl_ready is not None:
                    hl = hl_ready
                else:
                    hl = torch.empty((npix, Cc), dtype=T, device=dev)
                    ops.ln_forward(xin, None, hl, npix, Hc * Wc, Cc, 0, LN_EPS, self.ln_unbiased, dt)
                Hl, Wl = Hc, Wc
                Hc, Wc = Hc * 2, Wc * 2
                rec_t = lay.convs["unet." + lv.tail_key]
                # Upsample(nearest, x2) is never materialised (model/nn.py:184): the halo-patch kernels fetch patch pixel (ih, iw) from
                # (ih >> 1, iw >> 1) of the low-resolution map (conv and weight gradient alike); grids they do not tile go to the gather
                # kernel, which folds the upsampling into its per-tap gather.
                nxt = lay.levels[i - 1].ascent[0] if lay.levels[i - 1].ascent else None
                if nxt is not None and nxt.kind == "res":  # the up-conv also emits the next level's first LayerNorm input
                    cur, g_t, r_t, h0_carry = conv3("unet." + lv.tail_key, hl, Hl, Wl, Hc, Wc, CONV_UP, res=skips.pop(), want_ln=mod_of(nxt))
                else:
                    cur, g_t, r_t = conv3("unet." + lv.tail_key, hl, Hl, Wl, Hc, Wc, CONV_UP, res=skips.pop())
                if train:
                    def bw_tail(gy, xin=xin, hl=hl, g=g_t, rec=r_t, Hl=Hl, Wl=Wl, Hu=Hc, Wu=Wc, Cc=Cc, lvl=i - 1):
                        self.flush_wgrad_groups()  # the ascent side of the level above is complete
                        tape.gskip[lvl] = gy  # the skip operand receives the same gradient
                        self._wg(hl, gy, rec, g, dt)
                        # gradient w.r.t. the low-resolution map = 2x2 sums of the gradient w.r.t. its upsampling (adjoint of Upsample):
                        # summed in the input-gradient kernel's epilogue where it supports that -- the full-resolution gradient is
                        # then never written (537 MB at the top level) -- else a pooling pass behind it
                        gd = self._geom(B, Hu, Wu, rec.dg_ld, Hu, Wu, Cc, Cc, rec.cin, CONV_S1)
                        gl = torch.empty((B * Hl * Wl, Cc), dtype=T, device=dev)
                        if ops.conv_pool2_supported(gd, dt):
                            wop, wpk = self._conv_weights("d", rec, dt, gd)
                            ops.conv(gy, wop, None, gl, gd, dt, pool2=True, wpacked=wpk)
                        else:
                            gu = dgrad(rec, gy, Hu, Wu, Hu, Wu, CONV_S1, Cc)
                            ops.sumpool2(gu, gl, B, Hl, Wl, Cc, dt)
                        dx = torch.empty_like(gl)
                        ops.ln_backward(gl, xin, None, None, dx, None, B * Hl * Wl, Hl * Wl, Cc, 0, LN_EPS, self.ln_unbiased, dt)
                        tape.done(rec.w_off)
                        return dx
                    tape.steps.append(bw_tail)
            else:
                xin = cur
                if fold is not None and not train and nhwc_out and self._fold_output(fold, "unet." + lv.tail_key, xin, B, Hc, Wc, dt):
                    return None
                lfuse = None
                if loss_rows is not None:
                    lfuse = dict(sum=loss["sum"], gscale=loss["gscale"], scaler=loss.get("scaler"), eps=loss_rows[0], lde=loss_rows[1], C=lay.out_channels)
                cur, g_t, r_t = conv3("unet." + lv.tail_key, xin, Hc, Wc, Hc, Wc, CONV_S1, ldy=lay.cout_pad, cout=lay.cout_pad, loss=lfuse)
                if train:
                    tape.meta["loss_fused"] = lfuse is not None
                if train:
                    def bw_tail0(gy, xin=xin, g=g_t, rec=r_t, Hc=Hc, Wc=Wc, Cc=lv.channels):
                        gw = dict(g)
                        gw["Cout"] = rec.rows
                        self._wg(xin, gy, rec, gw, dt)
                        dxt = dgrad(rec, gy, Hc, Wc, Hc, Wc, CONV_S1, Cc)
                        tape.done(rec.w_off)
                        return dxt
                    tape.steps.append(bw_tail0)
        if train:
            tape.meta.update(B=B, C=C, H=H, W=W, dt=dt, out_nhwc=cur, ldm=ldm, Bt=Bt)
        if nhwc_out:
            return cur
        y = torch.empty((B, C, H, W), dtype=torch.float32, device=dev)
        ops.nhwc_to_nchw(cur, y, B, lay.out_channels, H * W, lay.cout_pad, dt)
        return y

    use_center_conv = os.environ.get("C2W_NO_CENTER_CONV") != "1"  # A/B knob
    # Opt-in (C2W_LN_CHAIN=1 or the attribute): the chain form of a level side -- residual-block outputs that only the next block reads are
    # not written, the next block rebuilds its residual from the LayerNorm rows this one emitted (res_block).  Measured at B = 128, bf16
    # (profiles/r06_experiments.md): -0.23 ms per step (five 128-channel launches lose their 537 / 134 MB store), but the rebuilt
    # residual carries the rounding of h = LN(x + m) scaled by sigma -- 2^-9 |x + m - mean| instead of 2^-9 |x| -- and where the modulation
    # dominates the block input that is MORE than rounding x itself: against the CPU oracle the worst gradient tensor moves from
    # 8.7e-3 to 1.7e-2 relative L2 (bf16; fp16 5e-3 -> 8.5e-3 of the output scale).  Parity before 0.5 %: off.
    chain_blocks = os.environ.get("C2W_LN_CHAIN") == "1"
    use_splitk = os.environ.get("C2W_NO_SPLITK") is None  # A/B knob: under-filled inference convs as one workgroup per tile (rounds 1-5)
    fuse_loss = os.environ.get("C2W_NO_LOSS_FUSION") is None  # A/B knob: the loss tail as its own pass (rounds 1-5); the library reads the same variable
    use_gemv = os.environ.get("C2W_NO_GEMV") != "1"  # A/B knob: one-row Linear layers as matrix-vector products

    def _fold_output(self, fold: dict, name: str, xin: torch.Tensor, B: int, H: int, W: int, dt: int) -> bool:
        """The output convolution of a batch of trajectory windows, restricted to the frames fold() keeps (see forward).  False: not
        available for this shape / type -- nothing was written."""
        lay = self.layout
        rec = lay.convs[name]
        k, F, nwin = int(fold["k"]), int(fold["F"]), int(fold["nwin"])
        w = 2 * k + 1
        if not self.use_center_conv or rec.rows != w * F or rec.kstride != rec.cin or \
                not ops.conv_center_supported(H, W, rec.cin, F, dt):
            return False
        HW = H * W
        wts, bias = self._w(rec, dt), self._b(rec)
        T = TORCH_DTYPE[dt]
        for eps, i0, nw, pos in fold["segs"]:
            rows = xin[pos * HW:]
            ops.conv_center(rows, wts, bias, eps[i0 + k:], nw, H, W, rec.cin, rec.rows, k * F, F, F * HW, dt)
            for gi in sorted({0, nwin - 1}):  # a trajectory's first / last window keeps k more frames: the full convolution of that one window
                if not i0 <= gi < i0 + nw:
                    continue
                one = xin[(pos + gi - i0) * HW: (pos + gi - i0 + 1) * HW]
                g = self._geom(1, H, W, rec.kstride, H, W, lay.cout_pad, lay.cout_pad, rec.rows, CONV_S1)
                wop, wpk = self._conv_weights("f", rec, dt, g)
                y1 = torch.empty((HW, lay.cout_pad), dtype=T, device=xin.device)
                ops.conv(one, wop, bias, y1, g, dt, wpacked=wpk)
                ops.window_scatter(y1, eps, 1, F, HW, k, gi, nwin, lay.cout_pad, dt)
        return True

    def _mlp_layer(self, name: str, x: torch.Tensor, rows: int, tape: Optional[Tape], need_dx: bool = True,
                   add: Optional[torch.Tensor] = None, add_bw: Optional[Callable] = None) -> torch.Tensor:
        """silu(Linear(x) [+ add]) of the time-embedding MLP (model/score.py:62-67; ``add`` = the forcing projection, :65-66); keeps the
        pre-activation when taping."""
        if tape is None and add is None:
            return self._linear(name, x, rows, ACT_SILU, None)
        z = self._linear(name, x, rows, ACT_NONE, None)
        if add is not None:
            z.add_(add)  # (rows, E) fp32: the one tensor-arithmetic line of the forward; its adjoint hands gz to both Linears unchanged
        if tape is None:
            h = torch.empty_like(z)
            ops.silu(z, h, z.numel(), DTYPE_F32)
            return h
        h = torch.empty_like(z)
        ops.silu(z, h, z.numel(), DTYPE_F32)
        rec = self.layout.convs[name]
        g = self._geom(rows, 1, 1, rec.kstride, 1, 1, rec.rows, rec.rows, rec.rows, CONV_1X1)

        def bw(gh: torch.Tensor) -> Optional[torch.Tensor]:
            gz = torch.empty_like(z)
            ops.silu_backward(z, gh, gz, z.numel(), DTYPE_F32)
            if add_bw is not None:  # z = Linear(x) + add: the same gz is the output gradient of whatever produced `add`
                add_bw(gz)
            self._wg(x, gz, rec, g, DTYPE_F32)
            if not need_dx:
                tape.done(rec.w_off)
                return None
            dx = torch.empty((rows, rec.cin), dtype=torch.float32, device=x.device)
            ops.conv(gz, self._wT(rec, DTYPE_F32), None, dx, self._geom(rows, 1, 1, rec.dg_ld, 1, 1, rec.cin, rec.cin, rec.cin, CONV_1X1),
                     DTYPE_F32)
            tape.done(rec.w_off)
            return dx
        tape.steps.append(bw)
        return h

    # ------------------------------------------------------------------ backward
    def backward(self, tape: Tape, gy_nhwc: torch.Tensor, want_dx: bool = False, want_dw: bool = True) -> Optional[torch.Tensor]:
        """Run the recorded closures in reverse.  ``gy_nhwc``: gradient w.r.t. the network output in NHWC rows
        [B*H*W][cout_pad] (padding channels zero).  Parameter gradients are ACCUMULATED into ``flat_grad``.
        ``want_dw=False``: input gradient only -- no weight-gradient launch, no modulation path (exact guidance through a network
        whose parameters do not require gradients, src/thor/score.py:28-33 on the frozen copy exp/downscaling.py:110-126 loads)."""
        if not want_dw:
            self._skip_dw = True
            try:
                return self._backward(tape, gy_nhwc, want_dx, False)
            finally:
                self._skip_dw = False
        if self.flat_grad is None:
            raise RuntimeError("call ensure_grad_buffer() before backward")
        return self._backward(tape, gy_nhwc, want_dx, True)

    def _backward(self, tape: Tape, gy_nhwc: torch.Tensor, want_dx: bool, want_dw: bool) -> Optional[torch.Tensor]:
        it = self.backward_steps(tape, gy_nhwc, want_dx, want_dw)
        try:
            while True:
                next(it)
        except StopIteration as done:
            return done.value

    def backward_steps(self, tape: Tape, gy_nhwc: torch.Tensor, want_dx: bool = False, want_dw: bool = True):
        """``backward`` as a generator: yields after every recorded closure the lowest flat offset whose gradient is final so far
        (``layout.numel`` before the first; see Tape.progress) -- a caller may hand finished suffixes of the gradient buffer on while
        the rest of the pass is still to be enqueued (score.py::_GradSegment: torch's DistributedDataParallel all-reduces a bucket as
        soon as its gradients have been delivered).  The generator's return value is dx (or None)."""
        low = [self.layout.numel]
        outer = tape.progress
        held = [None]

        def emit(off: int) -> None:
            low[0] = min(low[0], off)
            if outer is not None:
                outer(off)

        def progress(off: int) -> None:
            if self._wg_groups:  # a weight gradient at or above ``off`` is still queued (grouped launches): not final yet
                held[0] = off if held[0] is None else min(held[0], off)
                return
            emit(off)

        def release() -> None:
            if held[0] is not None and not self._wg_groups:
                off, held[0] = held[0], None
                emit(off)
        # Grouped weight gradients queued by ANOTHER backward that is still being consumed on this engine (two recorded forwards whose
        # backward generators interleave: score.py::_GradSegment chains driven by autograd) are legitimate pending work: launch them and
        # let that backward hand on the offsets it held, instead of dropping them silently (round-5 advisor finding).  What a backward
        # that RAISED left behind is dropped by its own generator (the finally clause below), never seen here.
        if self._wg_groups:
            self.flush_wgrad_groups()
        tape.progress = progress
        self._done_releases.append(release)
        completed = False
        try:
            dx0, m = yield from self._backward_body(tape, gy_nhwc, want_dx, want_dw, low, outer)
            completed = True
        finally:
            if release in self._done_releases:
                self._done_releases.remove(release)
            if not completed:  # raised or closed half-way: queued launches hold tensors of a pass that will never finish
                self._wg_groups.clear()
                tape.progress = outer
        if not want_dx or dx0 is None:
            return None
        dx = torch.empty((m["B"], m["C"], m["H"], m["W"]), dtype=torch.float32, device=dx0.device)
        ops.nhwc_to_nchw(dx0, dx, m["B"], m["C"], m["H"] * m["W"], self.layout.cin_pad, m["dt"])
        return dx

    def _backward_body(self, tape: Tape, gy_nhwc: torch.Tensor, want_dx: bool, want_dw: bool, low, outer):
        if self._dg_ready is not None:  # operands prefetched on the gradient stream: this stream reads them from here on
            torch.cuda.current_stream().wait_event(self._dg_ready)
            self._dg_ready = None
        # Build the input-gradient operands (transposed copies of ALL weights, read from the flat buffer) before the first launch on
        # the gradient stream: whoever updates a finished part of the flat buffer from that stream while backward is still running
        # (Trainer: optimizer chasing the backward) is then ordered behind these reads by the stream's first wait on this one.
        for grp_lin in (False, True):
            rec0 = next((r for r in self.layout.convs.values() if r.dg_off >= 0 and r.lin == grp_lin), None)
            if rec0 is not None:
                self._wT(rec0, DTYPE_F32 if grp_lin else tape.meta["dt"])
        steps = tape.steps
        n_mlp = 3  # map_layer0, map_layer1, proj were recorded first
        g = gy_nhwc
        for bw in reversed(steps[n_mlp:]):
            g = bw(g)
            yield low[0]
        dx0 = g
        self.flush_wgrad_groups()
        # modulation path: dm_all -> proj -> map_layer1 -> map_layer0 (parameter gradients only: t carries none)
        if want_dw:
            gm = tape.meta["dm_all"]
            for bw in reversed(steps[:n_mlp]):
                gm = bw(gm)
                yield low[0]
        self.join_grad_stream()  # every gradient is in flat_grad for whoever runs next on this stream (optimizer, autograd)
        tape.steps = []
        tape.gskip.clear()
        tape.progress = outer
        return dx0, tape.meta


def _module_links(net, names):
    """(parent module, child key, child module) for every module on the path to one of the parameters ``names``, each link once."""
    links, seen = [], set()
    for name in names:
        mod = net
        for p in name.split(".")[:-1]:
            child = mod._modules[p]
            if (id(mod), p) not in seen:
                seen.add((id(mod), p))
                links.append((mod, p, child))
            mod = child
    return links


def _resolve(net, name: str):
    parts = name.split(".")
    mod = net
    for p in parts[:-1]:
        mod = getattr(mod, p) if not p.isdigit() else mod[int(p)]
    return mod, parts[-1]
