"""The drop-in path on the GPU: ``climate2weather_amd.optim.AdamW`` (the ``optimizer_kwargs.class_name`` seam, train.py:175-180,
training_loop.py:119-123,380-384), ``SDAPipeline.loss`` as one autograd node (src/thor/pipelines.py:27-35) and the new kernels under
them (c2w_sq_err / c2w_sq_err_noise), through the C ABI.  The reference-shaped loop itself is timed by bench.py (``module_api``)."""
import io
import math

import numpy as np
import pytest
import torch

from _module_loop import _batch, _close, _loop
from climate2weather_amd import ops
from climate2weather_amd.ema import StandardEMA
from climate2weather_amd.ops import DTYPE_BF16, DTYPE_F16, DTYPE_F32, TORCH_DTYPE
from climate2weather_amd.optim import AdamW
from climate2weather_amd.pipelines import SDAPipeline
from climate2weather_amd.score import ScoreUNet, _LossTensor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TINY = dict(embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3, padding_mode="zeros")
HP = dict(lr=1e-3, weight_decay=1e-3, betas=[0.9, 0.999])


def _tiny(seed=3, precision="fp32"):
    torch.manual_seed(seed)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **TINY).to(DEV)
    net.precision = precision
    return net


@pytest.mark.parametrize("dt", [DTYPE_F32, DTYPE_BF16, DTYPE_F16])
@pytest.mark.parametrize("B,C,HW,ldc", [(2, 6, 256, 64), (3, 65, 1024, 128), (1, 52, 64, 64), (2, 80, 4096, 128)])
def test_sq_err_kernels(dt, B, C, HW, ldc):
    """out = (y - eps)^2 in NCHW fp32 from NHWC rows, eps read or regenerated; loss_sum = its sum.  Against torch on the same values
    (the subtraction and the square are exact fp32 operations on the stored values: bit-equal)."""
    if dt == DTYPE_F32 and ldc % 32:
        pytest.skip("fp32 rows come in 32-channel chunks")
    g = torch.Generator().manual_seed(B * C)
    y = torch.randn(B * HW, ldc, generator=g).to(TORCH_DTYPE[dt]).to(DEV)
    eps = torch.randn(B, C, HW, generator=g).to(DEV)
    want = (y[:, :C].float().view(B, HW, C).permute(0, 2, 1) - eps) ** 2
    out = torch.full((B, C, HW), float("nan"), device=DEV)
    ls = torch.zeros(1, device=DEV)
    assert ops.sq_err(y, eps, out, ls, B, C, HW, ldc, dt)
    assert torch.equal(out, want)
    assert ls.item() == pytest.approx(want.double().sum().item(), rel=1e-5)
    seed = 0x1234567 + C
    e2 = torch.empty(B, C, HW, device=DEV)
    ops.philox_normal(e2, e2.numel(), seed)
    out2 = torch.empty_like(out)
    ls2 = torch.zeros(1, device=DEV)
    assert ops.sq_err(y, seed, out2, ls2, B, C, HW, ldc, dt)
    want2 = (y[:, :C].float().view(B, HW, C).permute(0, 2, 1) - e2) ** 2
    assert torch.equal(out2, want2)
    assert ls2.item() == pytest.approx(want2.double().sum().item(), rel=1e-5)
    assert not ops.sq_err(y, seed, out2[:, :, : HW - 2].contiguous(), None, B, C, HW - 2, ldc, dt)  # HW % 4 != 0: caller's fallback


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("bf16", 2e-2), ("fp16", 3e-3)])
def test_fused_loss_equals_the_reference_arithmetic(precision, tol):
    """SDAPipeline.loss as ONE node (regenerated eps: neither eps nor x_t nor eps_pred exists as a tensor) against the reference's
    own composition (src/thor/pipelines.py:27-35) on the same draws -- t from torch's device generator, eps = the stream of the seed the
    fused path drew, materialised: the unreduced tensor, .mean() and every parameter gradient; then a non-mean reduction (general
    gradient route) and no_grad."""
    fused, plain = SDAPipeline(), SDAPipeline()
    plain.fused_loss = False
    a, b = _tiny(precision=precision), _tiny(precision=precision)
    x = _batch(0, DEV)
    w = torch.rand(2, 6, 16, 16, generator=torch.Generator().manual_seed(1)).to(DEV)
    for reduce_ in (lambda l: l.mean(), lambda l: (l * w).sum() / 7.0):
        a.zero_grad()
        b.zero_grad()
        torch.manual_seed(11)
        la = fused.loss(net=a, x=x)
        assert isinstance(la, _LossTensor)
        # the same draws, by hand: t (device generator), then the seed (CPU generator)
        torch.manual_seed(11)
        t = torch.rand(2, 1, 1, 1, dtype=torch.float32, device=DEV)
        seed = int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64).item())
        eps = torch.empty(2, 6, 16, 16, device=DEV)
        ops.philox_normal(eps, eps.numel(), seed)
        lb = (b(plain.mu(t) * x + plain.sigma(t) * eps, t) - eps) ** 2
        assert (la - lb).abs().max().item() <= tol * lb.abs().max().item()
        ra, rb = reduce_(la), reduce_(lb)
        assert ra.item() == pytest.approx(rb.item(), rel=tol)
        ra.backward()
        rb.backward()
        for (n, p), q in zip(a.named_parameters(), b.parameters()):
            assert (p.grad - q.grad).abs().max().item() <= 2 * tol * q.grad.abs().max().item() + 1e-12, n
    with torch.no_grad():
        torch.manual_seed(11)
        l0 = fused.loss(net=a, x=x)
    assert l0.grad_fn is None and l0.mean().item() == pytest.approx(lb.mean().item(), rel=tol)


@pytest.mark.parametrize("precision,autocast", [("fp32", None), ("auto", torch.bfloat16)])
def test_drop_in_adamw_equals_torch_adamw_on_the_same_gradients(precision, autocast):
    """N steps of the reference-shaped loop with the drop-in optimizer (one fused launch over the flat buffer, 16-bit shadow written by
    the same kernel) == torch.optim.AdamW stepped on the same gradients, to fp32 round-off; the lr the loop writes into param_groups
    is honoured; a checkpoint written by one loads into the other and training continues identically."""
    pipe = SDAPipeline()
    a, b = _tiny(precision=precision), _tiny(precision=precision)
    oa, ob = AdamW(a.parameters(), **HP), torch.optim.AdamW(b.parameters(), **HP)
    ema = StandardEMA(a, rates=[0.9])
    lr_fn = lambda i: 1e-3 * (1 - i / 10)  # noqa: E731
    _loop(a, oa, pipe, 4, lr_fn, mirror=(b, ob), dev=DEV, autocast=autocast)
    ema.update()
    assert 0 in oa._flat, "the flat path did not engage"
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), n
        sa, sb = oa.state[p], ob.state[q]
        assert float(sa["step"]) == float(sb["step"]) == 4
        assert _close(sa["exp_avg"], sb["exp_avg"]) and _close(sa["exp_avg_sq"], sb["exp_avg_sq"]), n
    # checkpoints both ways (src/thor/checkpoint.py:13-57), through the file format
    buf = io.BytesIO()
    torch.save(oa.state_dict(), buf)
    buf.seek(0)
    c, d = _tiny(seed=9, precision=precision), _tiny(seed=9, precision=precision)
    c.load_state_dict(a.state_dict())
    d.load_state_dict(b.state_dict())
    oc, od_ = torch.optim.AdamW(c.parameters(), **HP), AdamW(d.parameters(), **HP)
    oc.load_state_dict(torch.load(buf, weights_only=False))
    od_.load_state_dict(ob.state_dict())
    for i in (4, 5):
        _loop(a, oa, pipe, 1, lr_fn, mirror=(c, oc), first=i, dev=DEV, autocast=autocast)
        for p, r in zip(a.parameters(), d.parameters()):
            r.grad = p.grad.detach().clone()
        for g in od_.param_groups:
            g["lr"] = lr_fn(i)
        od_.step()
    for (n, p), q, r in zip(a.named_parameters(), c.parameters(), d.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), n
        assert torch.allclose(p, r, rtol=1e-5, atol=1e-7), n
    assert oa.steps_taken() == od_.steps_taken() == 6
    # the network the optimizer just wrote evaluates like a fresh one holding the same weights (shadow refreshed by the step kernel)
    fresh = _tiny(seed=1, precision=precision)
    fresh.load_state_dict(a.state_dict())
    x, t = _batch(9, DEV), torch.rand(2, device=DEV)
    with torch.no_grad(), (torch.autocast("cuda", dtype=autocast) if autocast else torch.autocast("cuda", enabled=False)):
        assert torch.equal(a(x, t), fresh(x, t))


def test_fp16_autocast_with_torch_grad_scaler():
    """Fabric's "16-mixed" (train.py:98) = fp16 autocast + torch.amp.GradScaler around the loop of training_loop.py:369-391: the
    scaler hands grad_scale / found_inf to the drop-in optimizer (fused-optimizer protocol), which unscales and skips on the device.
    A clean step == torch.optim.AdamW on the unscaled gradients; an overflowing step changes nothing, is not counted, halves the scale."""
    pipe = SDAPipeline()
    a, b = _tiny(precision="auto"), _tiny(precision="auto")
    oa, ob = AdamW(a.parameters(), **HP), torch.optim.AdamW(b.parameters(), **HP)
    sc = torch.amp.GradScaler("cuda", init_scale=256.0, growth_interval=1000)
    losses = _loop(a, oa, pipe, 3, scaler=sc, mirror=(b, ob), dev=DEV, autocast=torch.float16)
    assert all(math.isfinite(v) for v in losses) and 0 in oa._flat and oa._flat[0]["amp"] is not None
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=2e-5, atol=1e-7), n
    before = [p.detach().clone() for p in a.parameters()]
    oa.zero_grad()
    with torch.autocast("cuda", dtype=torch.float16):
        loss = pipe.loss(net=a, x=_batch(7, DEV)).mean()
    sc.scale(loss).backward()
    list(a.parameters())[5].grad.view(-1)[3] = float("nan")
    sc.step(oa)
    sc.update()
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), before))
    assert oa.steps_taken() == 3 and sc.get_scale() == 128.0
    _loop(a, oa, pipe, 1, scaler=sc, mirror=(b, ob), first=3, dev=DEV, autocast=torch.float16)
    assert oa.steps_taken() == 4
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=2e-5, atol=1e-7), n
    assert float(oa.state_dict()["state"][0]["step"]) == 4.0


def test_loss_item_does_not_drain_the_stream():
    """training_loop.py:385 reads ``loss.detach().item()`` AFTER optimizer.step(): on a plain tensor that waits for the backward pass
    and the update although the value was final at the end of the forward.  score.py::_LossScalar answers item() / float() from a
    pinned host memory the value was published into behind its own producer (Engine.publish): the same bits, while a second of later
    work is still running."""
    import time
    from climate2weather_amd.score import _LossScalar
    pipe = SDAPipeline()
    net = _tiny(precision="auto")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = pipe.loss(net=net, x=_batch(0, DEV)).mean().mul(0.5)
    assert isinstance(loss, _LossScalar) and isinstance(loss.detach(), _LossScalar) and loss.requires_grad
    loss.backward()
    assert all(p.grad is not None for p in net.parameters())
    want = loss.detach().as_subclass(torch.Tensor).item()  # the ordinary, draining read
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2.0e9))  # ~1 s of later work on the stream (the loop's backward + optimizer step)
    t0 = time.perf_counter()
    got = loss.detach().item()
    took = time.perf_counter() - t0
    assert not torch.cuda.current_stream().query(), "the sleep kernel finished before item() returned: nothing was shown"
    assert got == want and float(loss) == want and took < 0.5, (got, want, took)
    torch.cuda.synchronize()
    # anything else is an ordinary tensor operation with an ordinary result
    assert type(loss + 1.0) is torch.Tensor and (loss + 1.0).item() == pytest.approx(want + 1.0, rel=1e-6)
    assert type(loss.detach().cpu()) is torch.Tensor
    # the unfused route hands back plain tensors
    pipe.fused_loss = False
    with torch.autocast("cuda", dtype=torch.bfloat16):
        plain = pipe.loss(net=net, x=_batch(0, DEV)).mean()
    assert type(plain) is torch.Tensor


@pytest.mark.parametrize("precision,autocast", [("fp32", None), ("auto", torch.bfloat16)])
def test_segmented_gradient_delivery_equals_the_single_node_on_the_gpu(precision, autocast):
    """score.py::_GradSegment on the HIP engine (two-stream backward, events between the slices): the chain of nodes hands the same
    gradients over as the single node -- the 3x3 / 1x1 weight gradients bit for bit (split-K sums in a fixed order), everything that
    goes through fp32 atomics (biases, modulation path) to round-off -- and the first of them before the pass has been enqueued."""
    pipe = SDAPipeline()
    a, b = _tiny(precision=precision), _tiny(precision=precision)
    b.grad_segments = 4
    launches, order = [], []
    real = ops.conv_wgrad
    ops.conv_wgrad = lambda *a_, **kw: (launches.append(1), real(*a_, **kw))[1]
    try:
        for net in (a, b):
            if net is b:
                for n, p in b.named_parameters():
                    p.register_hook(lambda g, n=n: order.append(len(launches)))
                del launches[:]
            torch.manual_seed(5)
            with (torch.autocast("cuda", dtype=autocast) if autocast else torch.autocast("cuda", enabled=False)):
                loss = pipe.loss(net=net, x=_batch(0, DEV)).mean()
            loss.backward()
        torch.cuda.synchronize()
    finally:
        ops.conv_wgrad = real
    assert len(order) == len(list(b.parameters())) and order[0] < len(launches) and len(set(order)) >= 3
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        if n.endswith("weight") and p.dim() >= 3:
            assert torch.equal(p.grad, q.grad), n
        else:
            assert _close(q.grad, p.grad, 1e-4 if autocast is None else 2e-2), n
    # the optimizer's flat path takes these gradients (views of one private buffer) like the single node's
    ob = AdamW(b.parameters(), **HP)
    ob.step()
    assert 0 in ob._flat


def test_full_size_network_one_reference_shaped_step():
    """The default network (configs/sda_unet.yml, C = 65, 128 x 128) through one iteration of the reference's loop with all five
    class_name seams pointing here, bf16 autocast, B = 4: flat path engaged, EMA moved, and the weights equal torch.optim.AdamW's on
    the same gradients."""
    cfg = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3, padding_mode="zeros",
               attention_levels=[4])
    torch.manual_seed(0)
    a = ScoreUNet(channels=65, spatial=2, activation=torch.nn.SiLU, **cfg).to(DEV)
    torch.manual_seed(0)
    b = ScoreUNet(channels=65, spatial=2, activation=torch.nn.SiLU, **cfg).to(DEV)
    oa, ob = AdamW(a.parameters(), lr=1e-4, weight_decay=1e-3, betas=[0.9, 0.999]), torch.optim.AdamW(b.parameters(), lr=1e-4, weight_decay=1e-3)
    ema = StandardEMA(a)
    pipe = SDAPipeline()
    losses = _loop(a, oa, pipe, 2, mirror=(b, ob), dev=DEV, shape=(4, 65, 128, 128), autocast=torch.bfloat16)
    ema.update(cur_ndata=8, batch_size=4)
    assert 0 in oa._flat and all(0.5 < v < 3.0 for v in losses), losses
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), n
    moved = sum(float((e.detach() - p.detach()).abs().max()) > 0 for e, p in zip(ema.emas[0].parameters(), a.parameters()))
    assert moved > 200
