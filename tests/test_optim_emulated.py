"""climate2weather_amd.optim.AdamW -- the drop-in for ``optimizer_kwargs.class_name = "torch.optim.AdamW"`` (train.py:175-180,
training_loop.py:119-123,380-384) -- and the fused module-path loss (SDAPipeline.loss as one autograd node), on the CPU with the HIP
launchers replaced by tests/emu_ops.py.  The GPU forms live in tests/test_gpu_module_api.py."""
import copy
import io
import os

import numpy as np
import pytest
import torch

import emu_ops
from climate2weather_amd import ops as c2w_ops
from climate2weather_amd.ema import StandardEMA
from climate2weather_amd.optim import AdamW
from climate2weather_amd.pipelines import SDAPipeline
from climate2weather_amd.score import ScoreUNet, _LossTensor

TINY = dict(embedding_dim=64, hidden_channels=[32, 64], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3, padding_mode="zeros")
HP = dict(lr=1e-3, weight_decay=1e-3, betas=[0.9, 0.999])  # train.py:176-181 passes betas as a list


@pytest.fixture()
def emu(monkeypatch):
    emu_ops.install(monkeypatch, c2w_ops)


def _tiny(seed=3):
    torch.manual_seed(seed)
    return ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **TINY)


from _module_loop import _batch, _close, _loop  # noqa: E402


def test_flat_adamw_equals_torch_adamw_over_a_training_loop(emu):
    """N steps with the drop-in optimizer (ONE fused launch over the flat buffer) == N steps of torch.optim.AdamW on the same
    gradients, to fp32 round-off: weights, exp_avg, exp_avg_sq, step; lr written into param_groups by the loop is honoured."""
    pipe = SDAPipeline()
    pipe.fused_loss = "eps"
    lr_fn = lambda i: 1e-3 * (1 - i / 10)  # noqa: E731
    a, b = _tiny(), _tiny()
    oa, ob = AdamW(a.parameters(), **HP), torch.optim.AdamW(b.parameters(), **HP)
    _loop(a, oa, pipe, 4, lr_fn, mirror=(b, ob))
    assert 0 in oa._flat, "the flat path did not engage"
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), n
        sa, sb = oa.state[p], ob.state[q]
        assert float(sa["step"]) == float(sb["step"]) == 4
        assert _close(sa["exp_avg"], sb["exp_avg"]) and _close(sa["exp_avg_sq"], sb["exp_avg_sq"]), n
    assert [g["lr"] for g in oa.param_groups] == [g["lr"] for g in ob.param_groups]
    assert list(oa.param_groups[0].keys()) == list(ob.param_groups[0].keys())  # a param_groups entry is interchangeable


def test_checkpoints_move_between_the_drop_in_and_torch_adamw(emu):
    """state_dict() has torch.optim.AdamW's layout: written by one, loaded by the other, training continues identically
    (src/thor/checkpoint.py:13-57 saves / restores ``optimizer`` through its state_dict)."""
    pipe = SDAPipeline()
    pipe.fused_loss = "eps"
    a, b = _tiny(), _tiny()
    oa, ob = AdamW(a.parameters(), **HP), torch.optim.AdamW(b.parameters(), **HP)
    _loop(a, oa, pipe, 2, mirror=(b, ob))
    sda, sdb = oa.state_dict(), ob.state_dict()
    assert sda["state"].keys() == sdb["state"].keys() and set(sda["state"][0]) == set(sdb["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert all(sda["state"][i]["exp_avg"].shape == sdb["state"][i]["exp_avg"].shape for i in sda["state"])
    buf = io.BytesIO()
    torch.save(sda, buf)  # a real round trip through the file format
    buf.seek(0)
    # drop-in -> torch, torch -> drop-in, on fresh networks holding the trained weights
    c, d = _tiny(seed=9), _tiny(seed=9)
    c.load_state_dict(a.state_dict())
    d.load_state_dict(b.state_dict())
    oc, od_ = torch.optim.AdamW(c.parameters(), **HP), AdamW(d.parameters(), **HP)
    oc.load_state_dict(torch.load(buf, weights_only=False))
    od_.load_state_dict(sdb)
    for p, q, r in zip(a.parameters(), c.parameters(), d.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7) and torch.allclose(p, r, rtol=1e-5, atol=1e-7)
    # one network differentiates, all three optimizers step on its gradients: original, loaded-into-torch, loaded-into-drop-in
    for i in (2, 3):
        _loop(a, oa, pipe, 1, mirror=(c, oc), first=i)
        for p, r in zip(a.parameters(), d.parameters()):
            r.grad = p.grad.detach().clone()
        od_.step()
    for p, q, r in zip(a.parameters(), c.parameters(), d.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7) and torch.allclose(p, r, rtol=1e-5, atol=1e-7)
    assert oa.steps_taken() == 4 and od_.steps_taken() == 4 and float(oc.state[next(iter(c.parameters()))]["step"]) == 4


def test_grad_scaler_protocol_unscales_and_skips_on_the_device(emu):
    """torch.amp.GradScaler (Fabric's "16-mixed", train.py:98) drives the optimizer through grad_scale / found_inf: a clean step equals
    the unscaled torch step; a step whose gradients overflowed changes nothing, is not counted, and halves the scale."""
    pipe = SDAPipeline()
    pipe.fused_loss = "eps"
    a, b = _tiny(), _tiny()
    oa, ob = AdamW(a.parameters(), **HP), torch.optim.AdamW(b.parameters(), **HP)
    sc = torch.amp.GradScaler("cpu", init_scale=1024.0, growth_interval=1000)
    _loop(a, oa, pipe, 2, scaler=sc, mirror=(b, ob))
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=2e-5, atol=1e-7)
    # overflow: one gradient becomes inf after backward -> the step must be skipped on every tensor
    before = [p.detach().clone() for p in a.parameters()]
    oa.zero_grad()
    torch.manual_seed(5)
    loss = pipe.loss(net=a, x=_batch(7)).mean()
    sc.scale(loss).backward()
    next(iter(a.parameters())).grad[0, 0, 0, 0] = float("inf")
    sc.step(oa)
    sc.update()
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), before))
    assert oa.steps_taken() == 2 and sc.get_scale() == 512.0
    assert not hasattr(oa, "grad_scale") and not hasattr(oa, "found_inf")
    # and training resumes: the next clean step is step 3 of torch's optimizer
    _loop(a, oa, pipe, 1, scaler=sc, mirror=(b, ob), first=2)
    assert oa.steps_taken() == 3
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=2e-5, atol=1e-7)
    assert float(oa.state_dict()["state"][0]["step"]) == 3.0


def test_optimizer_built_before_the_first_forward_and_the_16_bit_shadow_follows(emu):
    """training_loop.py:116-123 builds the optimizer right after the module (no forward yet on most ranks): the flat plan resolves at
    the first step.  In bf16 the fused step refreshes the 16-bit shadow itself: the next forward equals a fresh network's."""
    pipe = SDAPipeline()
    pipe.fused_loss = "eps"
    torch.manual_seed(3)
    cfg = dict(TINY, hidden_channels=[64, 128])
    a = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **cfg)
    a.precision = "bf16"
    oa = AdamW(a.parameters(), **HP)
    ema = StandardEMA(a, rates=[0.5])
    assert a.__dict__["_engine"] is None
    calls = []
    orig = c2w_ops.cast_f32
    _loop(a, oa, pipe, 1)
    ema.update()
    c2w_ops.cast_f32 = lambda *args: (calls.append(1), orig(*args))[1]
    try:
        torch.manual_seed(77)
        l1 = pipe.loss(net=a, x=_batch(3)).mean().item()
    finally:
        c2w_ops.cast_f32 = orig
    assert calls == [], "the optimizer kernel writes the bf16 shadow: no separate cast of the flat buffer after a step"
    fresh = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **cfg)
    fresh.load_state_dict(a.state_dict())
    fresh.precision = "bf16"
    torch.manual_seed(77)
    assert pipe.loss(net=fresh, x=_batch(3)).mean().item() == l1
    for p, e, p0 in zip(a.parameters(), ema.emas[0].parameters(), _tiny_like(cfg).parameters()):
        assert torch.allclose(e, 0.5 * p0 + 0.5 * p, atol=1e-7)


def _tiny_like(cfg):
    torch.manual_seed(3)
    return ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **cfg)


def test_foreign_parameters_and_partial_sets_take_the_per_tensor_path(emu):
    """Not an engine's full parameter set -> torch's functional AdamW per tensor, same numbers as torch.optim.AdamW."""
    torch.manual_seed(0)
    lin_a = torch.nn.Linear(5, 3)
    lin_b = copy.deepcopy(lin_a)
    oa, ob = AdamW(lin_a.parameters(), **HP), torch.optim.AdamW(lin_b.parameters(), **HP)
    for i in range(3):
        x = torch.randn(4, 5, generator=torch.Generator().manual_seed(i))
        for lin, o in ((lin_a, oa), (lin_b, ob)):
            o.zero_grad()
            lin(x).square().mean().backward()
            o.step()
    assert not oa._flat
    for p, q in zip(lin_a.parameters(), lin_b.parameters()):
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-8)
    # half of a network's parameters: no flat plan either, and the engine sees the new weights
    net, ref = _tiny(), _tiny()
    half = [p for i, p in enumerate(net.parameters()) if i % 2 == 0]
    half_ref = [p for i, p in enumerate(ref.parameters()) if i % 2 == 0]
    o1, o2 = AdamW(half, **HP), torch.optim.AdamW(half_ref, **HP)
    pipe = SDAPipeline()
    pipe.fused_loss = "eps"
    for i in range(2):
        o1.zero_grad()
        torch.manual_seed(i)
        pipe.loss(net=net, x=_batch(i)).mean().backward()
        for p, q in zip(half, half_ref):
            q.grad = p.grad.detach().clone()
        o1.step()
        o2.step()
        for p, q in zip(net.parameters(), ref.parameters()):  # the engine must have seen the per-tensor writes: next loss uses them
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-7)
    assert not o1._flat
    for p, q in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError):
        AdamW(lin_a.parameters(), amsgrad=True)


def test_gradients_outside_the_flat_buffer_keep_the_fused_step_and_nothing_is_reallocated(emu):
    """torch DDP with gradient_as_bucket_view=True (or any loop that leaves each ``p.grad`` a tensor of its own) hands the optimizer
    gradients that do not lie in one flat buffer: they are gathered by one multi-tensor copy and the step stays the fused one.  A step
    with a MISSING gradient goes per tensor on the same state and the next complete one re-enters the fused path -- the flat moment
    buffers are never deleted or reallocated (round-4 advice: they were, every step), and the numbers are torch.optim.AdamW's."""
    pipe = SDAPipeline()
    pipe.fused_loss = "eps"
    a, b = _tiny(), _tiny()
    oa, ob = AdamW(a.parameters(), **HP), torch.optim.AdamW(b.parameters(), **HP)
    skip_name = next(n for n, _ in a.named_parameters() if n.endswith("proj_out.bias"))
    ptrs = None
    for i in range(6):
        oa.zero_grad()
        torch.manual_seed(1000 + i)
        pipe.loss(net=a, x=_batch(i)).mean().backward()
        for (n, p), q in zip(a.named_parameters(), b.parameters()):
            g = p.grad.detach().clone()
            p.grad = g.clone() if i != 0 else p.grad  # from step 1 on: every gradient a foreign tensor ("bucket view")
            q.grad = g
            if i == 3 and n == skip_name:  # one step with a missing gradient: torch skips that parameter, so must we
                p.grad = None
                q.grad = None
        oa.step()
        ob.step()
        st = oa._flat[0]
        if ptrs is None:
            ptrs = (st["m"].data_ptr(), st["v"].data_ptr())
        assert (st["m"].data_ptr(), st["v"].data_ptr()) == ptrs, f"moment buffers reallocated at step {i}"
        assert oa.fused_path_active() == (i != 3), f"step {i}"
        for p in a.parameters():
            assert oa.state[p]["exp_avg"].untyped_storage().data_ptr() == st["m"].untyped_storage().data_ptr()
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        if n == skip_name:  # torch counts steps per parameter (this one is at 5), the fused path per group (6): bias corrections differ
            assert torch.allclose(p, q, rtol=0, atol=1e-3), n
            continue
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), n
        sa, sb = oa.state[p], ob.state[q]
        assert _close(sa["exp_avg"], sb["exp_avg"]) and _close(sa["exp_avg_sq"], sb["exp_avg_sq"]), n
    # the checkpoint reports the group's counter (the reference never skips a parameter: training_loop.py:369-391)
    sd = oa.state_dict()
    assert {float(v["step"]) for v in sd["state"].values()} == {6.0}
    assert oa.steps_taken() == 6


def test_fused_loss_is_the_reference_loss_and_any_downstream_use_works(emu):
    """SDAPipeline.loss through the one-node path == the reference's tensor arithmetic (src/thor/pipelines.py:27-35) on the same
    draws: the unreduced tensor, .mean() (answered from the kernel's sum), and every gradient -- also when the caller does something
    other than .mean() with it (weighted sum: the general gradient route), under no_grad, and behind a wrapper's ``.module``."""
    fused, plain = SDAPipeline(), SDAPipeline()
    assert fused.__dict__ == {"eta": 1e-3}  # the reference checkpoints pipeline.__dict__: the (class-level) switch must not be in it
    fused.fused_loss, plain.fused_loss = "eps", False
    a, b = _tiny(), _tiny()
    x = _batch(0)
    w = torch.rand(2, 6, 16, 16, generator=torch.Generator().manual_seed(1))
    for reduce_ in (lambda l: l.mean(), lambda l: (l * w).sum() / 7.0, lambda l: l.sum(dim=(1, 2, 3)).max()):
        a.zero_grad()
        b.zero_grad()
        torch.manual_seed(11)
        la = fused.loss(net=a, x=x)
        torch.manual_seed(11)
        lb = plain.loss(net=b, x=x)
        assert isinstance(la, _LossTensor) and not isinstance(la.exp(), _LossTensor) and la.shape == lb.shape
        assert torch.allclose(la, lb, rtol=1e-5, atol=1e-7)
        ra, rb = reduce_(la), reduce_(lb)
        assert type(ra) is torch.Tensor and ra.item() == pytest.approx(rb.item(), rel=1e-5)
        ra.backward()
        rb.backward()
        for (n, p), q in zip(a.named_parameters(), b.parameters()):
            assert torch.allclose(p.grad, q.grad, rtol=1e-4, atol=1e-6 * q.grad.abs().max().item() + 1e-12), n
    with torch.no_grad():
        torch.manual_seed(11)
        l0 = fused.loss(net=a, x=x)
        assert l0.grad_fn is None and l0.mean().item() == pytest.approx(lb.mean().item(), rel=1e-5)

    class Wrapper(torch.nn.Module):  # what fabric.setup_module / DDP hand to pipeline.loss: the call must go through the wrapper
        def __init__(self, module):
            super().__init__()
            self.module, self.calls = module, 0

        def forward(self, *args, **kwargs):
            self.calls += 1
            return self.module(*args, **kwargs)
    wnet = Wrapper(a)
    torch.manual_seed(11)
    lw = fused.loss(net=wnet, x=x)
    assert wnet.calls == 1 and isinstance(lw, _LossTensor) and torch.allclose(lw, lb, rtol=1e-5, atol=1e-7)
    assert "_loss_request" not in a.__dict__
