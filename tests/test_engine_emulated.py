"""CPU checks of the host engine (flat parameter layout, launch sequencing, hand-written backward tape) with the HIP
launchers replaced by the PyTorch test double tests/emu_ops.py.  The kernels themselves are checked on the GPU
(tests/test_gpu_*.py); this file makes sure that what they are asked to do adds up to the reference network."""
import json
import os

import numpy as np
import pytest
import torch

import emu_ops
from climate2weather_amd import ops as c2w_ops
from climate2weather_amd.score import ScoreUNet
from oracle import diffusion as od
from oracle import unet as ou

TINY = dict(embedding_dim=64, hidden_channels=[32, 64], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3,
            padding_mode="zeros")


@pytest.fixture()
def emu(monkeypatch):
    emu_ops.install(monkeypatch, c2w_ops)


def _golden(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name), allow_pickle=False).items()}


def _tiny(seed=3, hidden_channels=None):
    torch.manual_seed(seed)
    cfg = dict(TINY, hidden_channels=hidden_channels) if hidden_channels else TINY
    return ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **cfg)


def test_creation_order_init_matches_reference(golden_dir):
    g = _golden(golden_dir, "tiny_net.npz")
    net = _tiny()
    sd = net.state_dict()
    ref = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    assert list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), ref[k]), k
    assert [n for n, _ in net.named_parameters()] == [str(n) for n in g["param_order"]]


def test_full_size_state_dict_fingerprint(golden_dir):
    fp = json.load(open(os.path.join(golden_dir, "full_net_fingerprint.json")))
    import yaml
    cfg = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3,
               padding_mode="zeros", attention_levels=[4])
    torch.manual_seed(0)
    net = ScoreUNet(channels=52, spatial=2, activation=torch.nn.SiLU, **cfg)
    sd = net.state_dict()
    assert list(sd.keys()) == fp["keys"]
    assert [list(v.shape) for v in sd.values()] == fp["shapes"]
    assert sum(p.numel() for p in net.parameters()) == fp["n_params"] == 72102964
    assert sd["unet.heads.0.weight"].double().sum().item() == pytest.approx(fp["heads0_sum"], rel=1e-12)
    assert sum(v.double().abs().sum().item() for v in sd.values()) == pytest.approx(fp["abs_sum"], rel=1e-12)


def test_flatten_keeps_values_and_views(emu, golden_dir):
    net = _tiny()
    before = {k: v.clone() for k, v in net.state_dict().items()}
    objs = {k: p for k, p in net.named_parameters()}
    strides = {k: p.stride() for k, p in net.named_parameters()}
    eng = net._get_engine()
    assert eng.is_attached(net)
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k]), k
    # attaching moves storage only: the Parameter OBJECTS (what optimizers / DDP reducers / EMA copies captured before the first
    # forward hold) and their strides (what DDP built its bucket views from) are unchanged, and every parameter is dense
    for k, p in net.named_parameters():
        assert p is objs[k] and p.stride() == strides[k], k
        assert p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last), k
        assert p.untyped_storage().data_ptr() == eng.flat.untyped_storage().data_ptr(), k
    # conv weights are K-contiguous [co][kh][kw][ci] from construction on
    w = net.unet.descent[0][0].residue[1].weight
    assert w.stride() == (9 * 32, 1, 3 * 32, 32)
    w0 = net.unet.heads[0].weight
    assert w0.shape == (32, 6, 3, 3) and w0.stride() == (9 * 6, 1, 3 * 6, 6)
    # deepcopy / state_dict round trip / pickle
    import copy, pickle
    net2 = copy.deepcopy(net)
    assert net2.__dict__["_engine"] is None
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k
    net3 = pickle.loads(pickle.dumps(net))
    net3.load_state_dict(net.state_dict())
    x = torch.randn(1, 6, 16, 16)
    t = torch.tensor([0.4])
    with torch.no_grad():
        assert torch.allclose(net(x, t), net2(x, t), atol=1e-6)
        assert torch.allclose(net(x, t), net3(x, t), atol=1e-6)


def test_forward_matches_golden(emu, golden_dir):
    g = _golden(golden_dir, "tiny_net.npz")
    net = _tiny().eval()
    with torch.no_grad():
        y = net(torch.from_numpy(g["xt"]), torch.from_numpy(g["t"]))
        y32 = net(torch.from_numpy(g["x32"]), torch.tensor(0.3))
    assert torch.allclose(y, torch.from_numpy(g["y"]), atol=2e-5)
    assert torch.allclose(y32, torch.from_numpy(g["y32"]), atol=2e-5)


@pytest.mark.parametrize("split_min_rows", [2048, 1])  # 1: the wide-Linear input-gradient route (wgrad kernel) for every Linear
def test_backward_matches_golden(emu, golden_dir, monkeypatch, split_min_rows):
    from climate2weather_amd.engine import Engine
    monkeypatch.setattr(Engine, "LINEAR_DGRAD_SPLIT_MIN_ROWS", split_min_rows)
    g = _golden(golden_dir, "tiny_net.npz")
    net = _tiny()
    x, t, eps = (torch.from_numpy(g[k]) for k in ("x", "t", "eps"))
    loss = od.loss(net, x, t, eps).mean()
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-5)
    loss.backward()
    for n, p in net.named_parameters():
        ref = torch.from_numpy(g["grad." + n])
        assert p.grad is not None, n
        assert torch.allclose(p.grad, ref, atol=1e-5 + 2e-4 * ref.abs().max().item()), (n, (p.grad - ref).abs().max().item())


def test_input_gradient_and_jacrev(emu, golden_dir):
    g = _golden(golden_dir, "tiny_net.npz")
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    net = _tiny()
    x = torch.from_numpy(g["x32"])[:2].clone().requires_grad_(True)
    t = torch.tensor(0.3)
    w = torch.randn(2, 6, 32, 32, generator=torch.Generator().manual_seed(0))
    (gx,) = torch.autograd.grad((net(x, t) * w).sum(), x)
    xo = x.detach().clone().requires_grad_(True)
    yo = ou.score_unet_forward(sd, xo, t, hidden_blocks=[1, 1], attention_levels=[1])
    (gxo,) = torch.autograd.grad((yo * w).sum(), xo)
    assert torch.allclose(gx, gxo, atol=1e-5 + 2e-4 * gxo.abs().max().item())
    # functorch path used by the guided score function (src/thor/score.py:28-33)
    def f(xx):
        return (net(xx, t) * w).sum()
    J = torch.func.jacrev(f, chunk_size=1)(x.detach())
    assert torch.allclose(J, gxo, atol=1e-5 + 2e-4 * gxo.abs().max().item())


def test_optimizer_created_before_first_forward_trains_the_engine_weights(emu, golden_dir):
    """training_loop.py:116-131 builds the optimizer (and DDP, EMA) from net.parameters() before any forward; the engine
    attaches lazily at the first forward.  The step must land in the weights the engine computes with."""
    g = _golden(golden_dir, "tiny_net.npz")
    net = _tiny()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3)
    x, t, eps = (torch.from_numpy(g[k]) for k in ("x", "t", "eps"))
    loss = od.loss(net, x, t, eps).mean()
    loss.backward()
    opt.step()
    from oracle import host as oh
    for n, p in net.named_parameters():
        p0, gr = torch.from_numpy(g["sd." + n]), torch.from_numpy(g["grad." + n])
        exp = oh.adamw_step(p0, gr, torch.zeros_like(p0), torch.zeros_like(p0), 1, 1e-3)[0]
        big = gr.abs() > 1e-5
        assert torch.allclose(p.detach()[big], exp[big], atol=3e-6), n
    with torch.no_grad():  # and the next forward uses the stepped weights (the engine sees the optimizer's in-place update)
        y1 = net(torch.from_numpy(g["xt"]), torch.from_numpy(g["t"]))
    assert (y1 - torch.from_numpy(g["y"])).abs().max().item() > 1e-5
    sd1 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    yo = ou.score_unet_forward(sd1, torch.from_numpy(g["xt"]), torch.from_numpy(g["t"]), hidden_blocks=[1, 1], attention_levels=[1])
    assert torch.allclose(y1, yo, atol=2e-5), (y1 - yo).abs().max().item()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("writer", ["sgd", "load_state_dict", "ema_reset"])
def test_weight_caches_follow_writes_through_the_parameters(emu, golden_dir, precision, writer):
    """Every cache derived from the weights (16-bit shadow, zero-padded input-conv operand -- 6 channels here, not a multiple
    of the K chunk --, input-gradient operands) must follow writes made through the Parameter objects: torch.optim steps,
    load_state_dict after the first forward, StandardEMA.reset.  Compared with a FRESH network holding the same weights:
    forward and every gradient."""
    from climate2weather_amd.ema import StandardEMA
    g = _golden(golden_dir, "tiny_net.npz")
    x, t, eps = (torch.from_numpy(g[k]) for k in ("x", "t", "eps"))
    hc = [64, 128] if precision == "bf16" else None  # the 16-bit kernels want hidden channels in whole 64-channel K chunks
    net = _tiny(hidden_channels=hc)
    net.precision = precision
    od.loss(net, x, t, eps).mean().backward()  # first forward + backward: every cache is built
    if writer == "sgd":
        opt = torch.optim.SGD(net.parameters(), lr=0.05)
        for _ in range(2):
            opt.step()
            opt.zero_grad()
            od.loss(net, x, t, eps).mean().backward()
        opt.step()
        target = net
    elif writer == "load_state_dict":
        other = _tiny(seed=11, hidden_channels=hc)
        net.load_state_dict(other.state_dict())
        target = net
    else:
        ema = StandardEMA(net, rates=[0.5])
        target = ema.emas[0]
        target.precision = precision
        with torch.no_grad():
            for p in target.parameters():
                p.mul_(0.5)
        od.loss(target, x, t, eps).mean().backward()  # caches of the copy built on the halved weights
        ema.reset()
    target.zero_grad()
    fresh = _tiny(seed=5, hidden_channels=hc)
    fresh.load_state_dict({k: v.detach().clone() for k, v in target.state_dict().items()})
    fresh.precision = precision
    la = od.loss(target, x, t, eps).mean()
    lb = od.loss(fresh, x, t, eps).mean()
    la.backward()
    lb.backward()
    assert la.item() == lb.item()
    for (n, a), (_, b) in zip(target.named_parameters(), fresh.named_parameters()):
        assert torch.equal(a.grad, b.grad), n


def test_packed_weight_copies_keep_their_addresses_and_follow_the_weights(emu, golden_dir, monkeypatch):
    """The stage-major packed copies (engine._packed; here the emulation's stand-in layout) are handed to launches above a size
    threshold.  (i) results equal the plain-operand run; (ii) a geometry that asks for more matrices later must not move the copies
    a captured hipGraph may still be reading -- the buffer and every offset stay where they were (round-3 advice: the buffer used to be
    reallocated); (iii) a weight update refreshes exactly the copies that are in use; (iv) prepare_forward packs nothing new."""
    g = _golden(golden_dir, "tiny_net.npz")
    x, t, eps = (torch.from_numpy(g[k]) for k in ("x", "t", "eps"))
    hc = [64, 128]

    def run(net, xx):
        net.zero_grad()
        loss = od.loss(net, xx, t, eps[:, :, : xx.shape[-2], : xx.shape[-1]]).mean()
        loss.backward()
        return loss.item(), [p.grad.clone() for p in net.parameters()]

    plain = _tiny(hidden_channels=hc)
    plain.precision = "bf16"
    ref16 = run(plain, x)
    net = _tiny(hidden_channels=hc)
    net.precision = "bf16"
    monkeypatch.setattr(emu_ops, "PACKED_FROM_PIXELS", 2 * 16 * 16)  # only the 16x16 level of a 16x16 input "takes packed weights"
    calls = []
    orig = c2w_ops.pack_conv_weights_batched
    monkeypatch.setattr(c2w_ops, "pack_conv_weights_batched", lambda src, dst, desc, n, dt: (calls.append(n), orig(src, dst, desc, n, dt)))
    out = run(net, x)
    assert out[0] == ref16[0] and all(torch.equal(a, b) for a, b in zip(out[1], ref16[1]))
    eng = net._get_engine()
    kf = ("f", c2w_ops.DTYPE_BF16)
    want0 = set(eng._pk_want[kf])
    assert want0 and all(".residue." in n or "tails" in n for n in want0) and len(want0) < sum(r.taps == 9 for r in eng.layout.convs.values())
    ptr0 = eng._pk[kf].data_ptr()
    offs0 = dict(eng._pk_tab[kf][0])
    n_calls = len(calls)
    eng.prepare_forward(c2w_ops.DTYPE_BF16)  # nothing changed: no launch, nothing new wanted
    assert len(calls) == n_calls and set(eng._pk_want[kf]) == want0
    # a larger input: the 32x32 -> 16x16 level now crosses the threshold too -> newcomers are packed, nothing moves
    x32 = torch.from_numpy(g["x32"])[:2]
    e32 = torch.randn(2, 6, 32, 32, generator=torch.Generator().manual_seed(0))
    net.zero_grad()
    l32 = od.loss(net, x32, t, e32).mean()
    plain.zero_grad()
    assert l32.item() == od.loss(plain, x32, t, e32).mean().item()
    assert set(eng._pk_want[kf]) > want0 and eng._pk[kf].data_ptr() == ptr0 and dict(eng._pk_tab[kf][0]) == offs0
    assert all(n <= len(eng._pk_want[kf]) - len(want0) or n <= len(eng._pk_want[("d", c2w_ops.DTYPE_BF16)]) for n in calls[n_calls:])
    # weights change through the Parameter objects: the copies in use are refreshed, results follow
    with torch.no_grad():
        for p, q in zip(net.parameters(), plain.parameters()):
            p.mul_(0.5)
            q.mul_(0.5)
    a, b = run(net, x), run(plain, x)
    assert a[0] == b[0] and all(torch.equal(u, v) for u, v in zip(a[1], b[1]))
    assert eng._pk[kf].data_ptr() == ptr0


def test_forcing_projection_matches_the_reference(emu, golden_dir):
    """forcing_dim > 0 (model/score.py:49-51,65-66): creation order (map_forcing first), forward with per-item and scalar t, loss and all
    gradients against the imported reference's (tests/golden/tiny_net_forcing.npz); state_dict round trip; the argument checks."""
    g = _golden(golden_dir, "tiny_net_forcing.npz")
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, forcing_dim=5, **TINY)
    assert list(net.state_dict().keys())[:2] == ["map_forcing.weight", "map_forcing.bias"] and len(net.state_dict()) == len(g["names"])
    assert sum(v.double().abs().sum().item() for v in net.state_dict().values()) == pytest.approx(float(g["sd_abs_sum"]), rel=1e-12)
    x, t, eps, forcing = (torch.from_numpy(g[k]) for k in ("x", "t", "eps", "forcing"))
    y = net(x, t, forcing=forcing)
    assert torch.allclose(y, torch.from_numpy(g["y"]), atol=2e-5)
    loss = ((y - eps) ** 2).mean()
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-5)
    loss.backward()
    named = dict(net.named_parameters())
    for n, ref in zip([str(v) for v in g["names"]], g["grad_norm"]):
        assert named[n].grad.double().norm().item() == pytest.approx(float(ref), rel=5e-4, abs=1e-9), n
    for n in ("map_forcing.weight", "map_forcing.bias", "map_layer1.weight", "map_layer0.bias", "unet.heads.0.weight"):
        ref = torch.from_numpy(g["grad." + n])
        assert torch.allclose(named[n].grad, ref, atol=1e-6 + 2e-4 * ref.abs().max().item()), n
    with torch.no_grad():
        assert torch.allclose(net(x[:1], torch.tensor(0.3), forcing=forcing[:1]), torch.from_numpy(g["y_scalar_t"]), atol=2e-5)
        with pytest.raises(ValueError):
            net(x, t)  # built with forcing_dim > 0: the vector is required
        with pytest.raises(AssertionError):
            _tiny()(x, t, forcing=forcing)  # model/score.py:60
    other = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, forcing_dim=5, **TINY)
    other.load_state_dict(net.state_dict())
    with torch.no_grad():
        assert torch.equal(other(x, t, forcing=forcing), net(x, t, forcing=forcing))


def test_fp16_snapshot_module_round_trip_runs(emu):
    """training_loop.py:254-265 pickles ``deepcopy(ema).cpu().eval().requires_grad_(False).to(torch.float16)`` and
    exp/downscaling.py:110-126 unpickles it and calls it: the half-precision module object itself must run (the engine keeps
    fp32 master weights: the first forward upcasts the fp16 values)."""
    import copy, pickle
    net = _tiny()
    x, t = torch.randn(1, 6, 16, 16), torch.tensor([0.3])
    with torch.no_grad():
        y0 = net(x, t)
    snap = copy.deepcopy(net).cpu().eval().requires_grad_(False).to(torch.float16)
    m = pickle.loads(pickle.dumps(dict(ema=snap)))["ema"]
    assert next(m.parameters()).dtype == torch.float16 and not m.training
    with torch.no_grad():
        y1 = m(x, t)
    assert y1.dtype == x.dtype
    assert (y1 - y0).abs().max().item() <= 5e-3 * y0.abs().max().item()  # fp16 rounding of the weights only


def test_relu_network_matches_the_reference(emu, golden_dir):
    """activation=torch.nn.ReLU -- the default of the reference's UNet (model/nn.py:118) -- through the engine: ACT_RELU / ACT_RELU_PAIR
    epilogues, the stored (a > 0) mask as the backward multiplier.  Forward, loss and every gradient against the imported reference."""
    g = _golden(golden_dir, "tiny_net.npz")
    r = _golden(golden_dir, "tiny_net_relu.npz")
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.ReLU, **TINY)
    assert net.unet.activation_kind == "relu"
    x, t, eps = (torch.from_numpy(g[k]) for k in ("x", "t", "eps"))
    with torch.no_grad():
        y32 = net(torch.from_numpy(g["x32"]), torch.tensor(0.3))
    assert torch.allclose(y32, torch.from_numpy(r["y32"]), atol=2e-5)
    loss = od.loss(net, x, t, eps).mean()
    assert loss.item() == pytest.approx(float(r["loss"]), rel=1e-5)
    loss.backward()
    for n, p in net.named_parameters():
        ref = torch.from_numpy(r["grad." + n])
        assert torch.allclose(p.grad, ref, atol=1e-5 + 2e-4 * ref.abs().max().item()), (n, (p.grad - ref).abs().max().item())
    with pytest.raises(NotImplementedError):
        ScoreUNet(channels=6, spatial=2, activation=torch.nn.GELU, **TINY)


def test_weight_gradients_of_a_level_go_out_together_and_done_waits_for_them(emu, golden_dir):
    """The residual-block convs of a level side share one geometry: their weight gradients are queued during the backward and launched
    as ONE grouped call at the level boundary (model/nn.py:146-159; ops.conv_wgrad_grouped).  A "done" offset (Tape.progress: the
    trainer may all-reduce and update everything at or above it) is handed on only once every weight gradient at or above it has
    been launched -- checked by comparing, at every notification, the finished suffix of the gradient buffer with its final value."""
    from climate2weather_amd.engine import Tape
    from climate2weather_amd.ops import DTYPE_F32
    net = _tiny()
    eng = net._get_engine()
    eng.ensure_grad_buffer()
    x = torch.randn(2, 6, 32, 32, generator=torch.Generator().manual_seed(5))
    t = torch.rand(2, generator=torch.Generator().manual_seed(6))

    def run(group):
        eng.group_wgrads = group
        eng.flat_grad.zero_()
        emu_ops.GROUPED_LAUNCHES.clear()
        tape = Tape()
        y = eng.forward(x, t, DTYPE_F32, tape=tape, nhwc_out=True)
        seen = []
        tape.progress = lambda off: seen.append((off, eng.flat_grad[off:].clone()))
        eng.backward(tape, torch.ones_like(y) / y.numel())
        return eng.flat_grad.clone(), seen, list(emu_ops.GROUPED_LAUNCHES)

    g1, seen1, launches1 = run(True)
    g0, seen0, launches0 = run(False)
    # ascent 0 | level 1, both sides: its two attention blocks' proj_out and qkv 1x1 weight gradients (side "1": a 1x1 conv's pixels are rows), then
    # the four 3x3 convs | descent 0
    assert launches0 == [] and launches1 == [(32, 2), (1, 2), (1, 2), (16, 4), (32, 2)], launches1
    assert torch.equal(g0, g1)  # the emulation adds the same numbers in the same order either way
    for seen, final in ((seen1, g1), (seen0, g0)):
        offs = [o for o, _ in seen]
        assert offs == sorted(offs, reverse=True) and offs[-1] == 0 or min(offs) >= 0
        for off, suffix in seen:
            assert torch.equal(suffix, final[off:]), f"gradients at or above offset {off} were still being written when it was reported done"
    assert len(seen1) < len(seen0)  # grouped: fewer, larger notifications


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_residual_blocks_whose_intermediate_outputs_are_never_written(emu, monkeypatch, precision):
    """Round 6, the chain form of a level side (model/nn.py:27-28,146-159; engine.res_block): block k's second conv emits the next block's
    normalised input with its mean and 1/sigma and does NOT write the block output; block k + 1 rebuilds its residual from them.  Against
    the written form (emu_ops.CHAIN off) on a three-block level: same loss and gradients up to the 16-bit rounding the written form
    applies to the intermediate outputs, and against the fp32 oracle inside the mode's usual tolerance; the launches are counted."""
    from climate2weather_amd.training import Trainer
    cfg = dict(embedding_dim=64, hidden_channels=[64, 64], hidden_blocks=[3, 2], attention_levels=[], kernel_size=3, padding_mode="zeros")
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(2, 6, 16, 16, generator=gen) * 0.5 + 0.5
    t, eps = torch.rand(2, generator=gen), torch.randn(2, 6, 16, 16, generator=gen)
    calls = []
    real_conv = c2w_ops.conv

    def counting(*a, **k):
        calls.append((bool(k.get("no_y")), k.get("resn") is not None, k.get("lnf") is not None and k["lnf"].get("mean") is not None))
        return real_conv(*a, **k)
    monkeypatch.setattr(c2w_ops, "conv", counting)
    res = {}
    for chain in (True, False):
        monkeypatch.setattr(emu_ops, "CHAIN", chain)
        calls.clear()
        torch.manual_seed(5)
        net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **cfg)
        tr = Trainer(net, precision=precision, ema_rates=())
        tr.eng.chain_blocks = True  # opt-in (engine.Engine.chain_blocks); the emulation's CHAIN switch stands for the kernels' existence
        tr.eng.flat_grad.zero_()
        loss = tr._forward_backward(x, t, eps, sync=False)
        S = tr.loss_scale()
        res[chain] = (float(loss), tr.eng.flat_grad.clone() / S, list(calls))
    no_y = [c for c in res[True][2] if c[0]]
    # an output is not written when the NEXT block of the side emits a LayerNorm too (a side's last block does so only in front of an
    # up-block: the ascent side of level 1): descent 0: 1 of 3, descent 1: 0 of 2, ascent 1: 1 of 2, ascent 0: 1 of 3 -- each with its
    # mean kept, each followed by a block that rebuilds its residual
    assert len(no_y) == 3 and all(c[2] for c in no_y) and sum(1 for c in res[True][2] if c[1]) == 3
    assert not any(c[0] or c[1] or c[2] for c in res[False][2])
    tol = 3e-2 if precision == "bf16" else 1e-2  # two 16-bit runs that round at different places (observed 6.3e-3 on a modulation projection in fp16)
    assert abs(res[True][0] - res[False][0]) <= tol * abs(res[False][0])
    ga, gb = res[True][1], res[False][1]
    assert not torch.equal(ga, gb)
    views = tr.eng.layout.views
    for name, (off, shape, strides) in views.items():
        n = int(np.prod(shape))
        a, b_ = ga[off:off + n], gb[off:off + n]
        assert (a - b_).abs().max().item() <= tol * max(b_.abs().max().item(), 1e-12), name
    # and against the fp32 oracle
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    yo = ou.score_unet_forward(sd, od.perturb(x, t.view(-1, 1, 1, 1), eps), t, cfg["hidden_blocks"], cfg["attention_levels"])
    lo = ((yo - eps) ** 2).mean()
    assert abs(res[True][0] - lo.item()) <= tol * lo.item()


def test_two_backward_passes_on_one_engine_may_interleave_and_one_may_fail(emu):
    """Grouped weight gradients are queued on the engine between level boundaries.  (a) A second backward generator that starts while
    another is half-consumed (two recorded forwards, their autograd chains interleaved) must not drop what the first one queued: both
    passes' gradients add up to the two passes run one after the other.  (b) A backward that raises half-way drops its own queue -- the
    next pass does not launch weight gradients of the dead one.  (Round-5 advisor finding: the queue used to be cleared at the start of
    every pass.)"""
    from climate2weather_amd.engine import Tape
    from climate2weather_amd.ops import DTYPE_F32
    net = _tiny()
    eng = net._get_engine()
    eng.ensure_grad_buffer()
    gen = torch.Generator().manual_seed(9)
    xs = [torch.randn(2, 6, 32, 32, generator=gen) for _ in range(2)]
    ts = [torch.rand(2, generator=gen) for _ in range(2)]

    def record(i):
        tape = Tape()
        y = eng.forward(xs[i], ts[i], DTYPE_F32, tape=tape, nhwc_out=True)
        return tape, torch.ones_like(y) / y.numel()
    # reference: one after the other
    eng.flat_grad.zero_()
    for i in range(2):
        tape, gy = record(i)
        eng.backward(tape, gy)
    ref = eng.flat_grad.clone()
    # (a) interleaved: A runs two closures (weight gradients queued, nothing launched), B runs to the end, A resumes
    eng.flat_grad.zero_()
    (ta, ga), (tb, gb) = record(0), record(1)
    ita = eng.backward_steps(ta, ga)
    for _ in range(2):  # the output conv, then a residual block: its two weight gradients are queued until the level boundary
        next(ita)
    assert eng._wg_groups  # something of pass A is queued
    for _ in eng.backward_steps(tb, gb):
        pass
    for _ in ita:
        pass
    assert not eng._wg_groups and not eng._done_releases
    assert torch.allclose(eng.flat_grad, ref, rtol=1e-5, atol=1e-7) and (eng.flat_grad - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()
    # (b) a pass that dies half-way leaves nothing behind
    eng.flat_grad.zero_()
    ta, ga = record(0)
    ita = eng.backward_steps(ta, ga)
    for _ in range(2):  # the output conv, then a residual block: its two weight gradients are queued until the level boundary
        next(ita)
    assert eng._wg_groups
    ita.close()  # what an exception inside the consumer does to the generator
    assert not eng._wg_groups and not eng._done_releases
    eng.flat_grad.zero_()
    tb, gb = record(1)
    eng.backward(tb, gb)
    only_b = eng.flat_grad.clone()
    eng.flat_grad.zero_()
    tb, gb = record(1)
    eng.backward(tb, gb)
    assert torch.equal(only_b, eng.flat_grad)
