"""The seam the other way round: the REFERENCE's own sampler code (src/thor/score.py, src/thor/pipelines.py, loaded by path from
/root/reference -- build container only, skipped elsewhere) driving THIS package's ScoreUNet, i.e. a user who changes nothing
but `network_kwargs.class_name`.  HIP launchers are replaced by tests/emu_ops.py (CPU)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

import emu_ops
from climate2weather_amd import ops as c2w_ops
from climate2weather_amd.score import ScoreUNet

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "thor")), reason="reference checkout not present")

TINY = dict(embedding_dim=64, hidden_channels=[32, 64], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3, padding_mode="zeros")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reference_sampler_drives_this_network(monkeypatch, golden_dir):
    emu_ops.install(monkeypatch, c2w_ops)
    ref_pipe = _load("ref_pipelines_seam", f"{REF}/src/thor/pipelines.py")
    ref_score = _load("ref_score_seam", f"{REF}/src/thor/score.py")
    s = {k: v for k, v in np.load(os.path.join(golden_dir, "sampler.npz"), allow_pickle=False).items()}
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **TINY).eval()
    pipe = ref_pipe.SDAPipeline()
    sf = ref_score.BatchedScoreFunction(net, markov_order=1, batch_size=4, device=torch.device("cpu"), noise_process=pipe)
    with torch.no_grad():
        y = sf(torch.from_numpy(s["score_x"]), torch.tensor(0.7))
    assert torch.allclose(y, torch.from_numpy(s["score_y"]), atol=2e-5)
    # unconditioned trajectory (no corrector draws needed)
    x = pipe.sample(sf, torch.from_numpy(s["uncond_c0.noise"]), steps=4, corrections=0, tau=0.5, show_progressbar=False)
    ref = torch.from_numpy(s["uncond_c0.x"])
    assert (x - ref).abs().max().item() <= 3e-4 * ref.abs().max().item()
    # guided trajectory: the reference's torch.func.jacrev wraps this network's forward (exact_grad False and True)
    A = lambda z: torch.nn.functional.avg_pool2d(z[::2], 8)  # noqa: E731
    for name, exact in (("cond_c0", False),):
        sfc = ref_score.BatchedScoreFunction(net, markov_order=1, batch_size=4, device=torch.device("cpu"), noise_process=pipe)
        sfc.condition_on(A=A, y=torch.from_numpy(s["y_obs"]), std=torch.from_numpy(s["std"]), gamma=float(s["gamma"]), exact_grad=exact)
        xc = pipe.sample(sfc, torch.from_numpy(s[name + ".noise"]), steps=4, corrections=0, tau=0.5, show_progressbar=False)
        refc = torch.from_numpy(s[name + ".x"])
        assert (xc - refc).abs().max().item() <= 3e-4 * refc.abs().max().item(), name
    # exact_grad=True differentiates through the network inside jacrev: one guided score evaluation against this package's own
    sfe = ref_score.BatchedScoreFunction(net, markov_order=1, batch_size=4, device=torch.device("cpu"), noise_process=pipe)
    sfe.condition_on(A=A, y=torch.from_numpy(s["y_obs"]), std=torch.from_numpy(s["std"]), gamma=float(s["gamma"]), exact_grad=True)
    from climate2weather_amd.pipelines import SDAPipeline
    from climate2weather_amd.score_fn import BatchedScoreFunction
    mine = BatchedScoreFunction(net, markov_order=1, batch_size=4, device=torch.device("cpu"), noise_process=SDAPipeline())
    mine.device_resident = False
    mine.condition_on(A=A, y=torch.from_numpy(s["y_obs"]), std=torch.from_numpy(s["std"]), gamma=float(s["gamma"]), exact_grad=True)
    xq = torch.from_numpy(s["score_x"])
    e_ref = sfe(xq, torch.tensor(0.7))
    e_mine = mine(xq, torch.tensor(0.7))
    assert torch.allclose(e_ref, e_mine, rtol=1e-4, atol=1e-5)


def test_reference_loss_ema_and_default_score_function_with_this_network(monkeypatch, golden_dir):
    """src/thor/pipelines.py::SDAPipeline.loss, src/thor/ema.py::StandardEMA (deep copies + zip over parameters()) and
    src/thor/score.py::DefaultScoreFunction from the reference, over this package's ScoreUNet."""
    emu_ops.install(monkeypatch, c2w_ops)
    ref_pipe = _load("ref_pipelines_seam2", f"{REF}/src/thor/pipelines.py")
    ref_score = _load("ref_score_seam2", f"{REF}/src/thor/score.py")
    ref_ema = _load("ref_ema_seam2", f"{REF}/src/thor/ema.py")
    g = {k: v for k, v in np.load(os.path.join(golden_dir, "tiny_net.npz"), allow_pickle=False).items()}
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **TINY)
    ema = ref_ema.StandardEMA(net, rates=[0.9, 0.999])
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-3)
    pipe = ref_pipe.SDAPipeline()
    x = torch.from_numpy(g["x"])
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    torch.manual_seed(0)
    losses = []
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss = pipe.loss(net, x).mean()  # draws t ~ U(0,1) and eps itself (src/thor/pipelines.py:27-35)
        loss.backward()
        opt.step()
        ema.update()
        losses.append(loss.item())
    assert all(np.isfinite(losses))
    for (rate, m) in zip(ema.rates, ema.emas):
        for n, p in m.named_parameters():
            assert not torch.equal(p, before[n]) or p.numel() == 0, n
    for m, tag in ema.get():
        assert isinstance(m, ScoreUNet) and tag.startswith("-0.")
    # the deep copies are independent networks with their own engines: same input, different (averaged) weights
    xt, t = torch.from_numpy(g["xt"]), torch.from_numpy(g["t"])
    with torch.no_grad():
        assert not torch.allclose(net(xt, t), ema.emas[0](xt, t), atol=1e-7)
    s = {k: v for k, v in np.load(os.path.join(golden_dir, "sampler.npz"), allow_pickle=False).items()}
    torch.manual_seed(3)
    fresh = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **TINY).eval()
    dsf = ref_score.DefaultScoreFunction(fresh, markov_order=1, noise_process=pipe)
    with torch.no_grad():
        y = dsf(torch.from_numpy(s["score_x"]), torch.tensor(0.7))
    assert torch.allclose(y, torch.from_numpy(s["score_y"]), atol=2e-5)


def test_member_random_stream_is_the_reference_drivers(monkeypatch):
    """exp/downscaling.py:100-103,248-265 restated by oracle/host.py::ensemble_members: the driver itself is not importable here
    (fire / xarray / lightning), but the two things it does with random numbers are -- seed the process, `torch.randn(L,C,H,W)` per
    member, then the REFERENCE's `pipeline.sample`, whose corrector fills `z.normal_()` from the same CPU stream.  Running those
    statements with the reference's sampler classes must give the members `run_ensemble(rng="reference")` gives."""
    emu_ops.install(monkeypatch, c2w_ops)
    from climate2weather_amd.sampling import run_ensemble
    from oracle import host as oh
    ref_pipe = _load("ref_pipelines_seam2", f"{REF}/src/thor/pipelines.py")
    ref_score = _load("ref_score_seam2", f"{REF}/src/thor/score.py")
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **TINY).eval()
    L, C, H, W, rank, world, n, seed = 5, 2, 16, 16, 1, 2, 4, 7
    mine = run_ensemble(net, world=world, rank=rank, device=torch.device("cpu"), precision="fp32", length=L, n_vars=C, height=H, width=W,
                        markov_order=1, num_samples=n, steps=3, corrections=1, tau=0.5, batch_size=2, seed=seed)
    pipe = ref_pipe.SDAPipeline()
    sf = ref_score.BatchedScoreFunction(net, markov_order=1, batch_size=2, device=torch.device("cpu"), noise_process=pipe)
    oh.seed_everything(hash((seed, rank)) % (1 << 31))  # util.py:27-29
    per = n // world
    for i in range(per):  # exp/downscaling.py:248-258
        noise_vec = torch.randn(L, C, H, W)
        x = pipe.sample(sf, noise_vec, steps=3, corrections=1, tau=0.5, show_progressbar=False)
        sid, xm = mine[i]
        assert sid == rank * per + i
        assert (xm - x).abs().max().item() <= 3e-4 * x.abs().max().item()


def test_the_training_loop_with_five_strings_changed_follows_the_reference(monkeypatch):
    """training_loop.py:85-131,369-391 executed twice from the same seed: once with the reference's own classes (model.score.ScoreUNet
    over the zuko shim, torch.optim.AdamW, thor.pipelines.SDAPipeline, thor.ema.StandardEMA, thor.lr) and once with the five class_name /
    func_name strings of train.py:164-193 pointing at this package -- everything resolved by name through construct_class_by_name the
    way the loop does it.  Same draws (CPU tensors take the reference's loss arithmetic), so after three optimizer steps the losses,
    the 228-key state_dict, the EMA copy and the optimizer's state_dict must agree."""
    import sys
    emu_ops.install(monkeypatch, c2w_ops)
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_shim")
    monkeypatch.syspath_prepend(REF)
    monkeypatch.syspath_prepend(shim)
    for name, path in (("thor_pipelines_ref", f"{REF}/src/thor/pipelines.py"), ("thor_ema_ref", f"{REF}/src/thor/ema.py"), ("thor_lr_ref", f"{REF}/src/thor/lr.py")):
        sys.modules[name] = _load(name, path)
    from climate2weather_amd import util as c2w_util

    def run(strings):
        net_kw = c2w_util.EasyDict(class_name=strings["net"], channels=6, spatial=2, activation=torch.nn.SiLU, **TINY)
        opt_kw = c2w_util.EasyDict(class_name=strings["opt"], lr=1e-3, weight_decay=1e-3, betas=[0.9, 0.999])  # train.py:175-180
        pipe_kw = c2w_util.EasyDict(class_name=strings["pipe"])
        ema_kw = c2w_util.EasyDict(class_name=strings["ema"])
        lr_kw = c2w_util.EasyDict(func_name=strings["lr"], ref_lr=1e-3, total_ndata=40)
        torch.manual_seed(42)
        net = c2w_util.construct_class_by_name(**net_kw)  # training_loop.py:88-89
        net.train()
        pipeline = c2w_util.construct_class_by_name(**pipe_kw)
        optimizer = c2w_util.construct_class_by_name(params=net.parameters(), **opt_kw)  # training_loop.py:119-121
        ema = c2w_util.construct_class_by_name(net=net, **ema_kw)
        state = c2w_util.EasyDict(cur_ndata=0)
        batch_size, losses = 2, []
        g = torch.Generator().manual_seed(7)
        for _ in range(3):  # training_loop.py:369-391
            optimizer.zero_grad()
            data = torch.randn(batch_size, 6, 16, 16, generator=g) * 0.5 + 0.5
            loss = pipeline.loss(net=net, x=data).mean().mul(1.0)
            loss.backward()
            lr = c2w_util.call_func_by_name(cur_ndata=state.cur_ndata, **lr_kw)
            for grp in optimizer.param_groups:
                grp["lr"] = lr
            optimizer.step()
            losses.append(loss.detach().item())
            state.cur_ndata += batch_size
            ema.update(cur_ndata=state.cur_ndata, batch_size=batch_size)
        return losses, net.state_dict(), ema.emas[0].state_dict(), optimizer.state_dict()

    ref = run(dict(net="model.score.ScoreUNet", opt="torch.optim.AdamW", pipe="thor_pipelines_ref.SDAPipeline", ema="thor_ema_ref.StandardEMA",
                   lr="thor_lr_ref.linear_learning_rate_schedule"))
    mine = run(dict(net="climate2weather_amd.score.ScoreUNet", opt="climate2weather_amd.optim.AdamW", pipe="climate2weather_amd.pipelines.SDAPipeline",
                    ema="climate2weather_amd.ema.StandardEMA", lr="climate2weather_amd.lr.linear_learning_rate_schedule"))
    assert np.allclose(ref[0], mine[0], rtol=2e-5), (ref[0], mine[0])
    keys = [k for k in ref[1] if not k.endswith(".eps")]  # zuko's LayerNorm registers eps as a buffer
    assert keys == list(mine[1].keys()) and len(keys) == len(list(mine[1]))
    # Two independent differentiations (torch autograd over the reference's modules / this package's hand-written backward) agree to
    # fp32 round-off; Adam turns the round-off of a vanishing gradient (the attention key bias has none at all; single entries
    # elsewhere) into a full step of lr = 1e-3 per iteration.  So: essentially every entry equal, none further apart than the three steps.
    for k in keys:
        for a, b in ((ref[1][k], mine[1][k]), (ref[2][k], mine[2][k])):
            d = (a - b).abs()
            assert d.max().item() <= 3 * 1e-3 * 1.05, (k, d.max().item())
            if not k.endswith("qkv.bias"):
                assert d.mean().item() <= 2e-6 and (d > 1e-5).float().mean().item() <= 0.02, (k, d.mean().item())
    so_r, so_m = ref[3], mine[3]
    assert so_r["state"].keys() == so_m["state"].keys() and all(float(so_m["state"][i]["step"]) == 3.0 for i in so_m["state"])
    assert so_r["param_groups"][0]["lr"] == so_m["param_groups"][0]["lr"]
