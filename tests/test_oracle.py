"""Pin the CPU oracle (oracle/) to the golden vectors generated from the imported reference
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import diffusion as od
from oracle import host as oh
from oracle import unet as ou

TINY = dict(hidden_blocks=[1, 1], attention_levels=[1])
TINY_CFG = dict(embedding_dim=64, hidden_channels=[32, 64], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3, padding_mode="zeros")


def _load(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name), allow_pickle=False).items()}


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_kat_embedding_and_schedule(golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "kat.json")))
    t = torch.tensor(kat["timestep_embedding_t"])
    assert torch.allclose(ou.timestep_embedding(t, 32), torch.tensor(kat["timestep_embedding"]), atol=1e-7)
    assert torch.allclose(od.mu(t), torch.tensor(kat["mu"]), atol=1e-7)
    assert torch.allclose(od.sigma(t), torch.tensor(kat["sigma"]), atol=1e-7)
    # SURVEY 8a KATs
    e = ou.timestep_embedding(torch.tensor([0.5]), 32)[0]
    assert abs(e[0].item() - 0.87758255) < 1e-6 and abs(e[16].item() - 0.47942555) < 1e-6
    assert abs(od.mu(torch.tensor(0.25)).item() - 0.85910004) < 1e-6
    assert abs(od.sigma(torch.tensor(0.5)).item() - 0.85670274) < 1e-6


def test_kat_host(golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "kat.json")))
    assert oh.seed_from_args(42, 0) == kat["seed_hash"]["42,0"] == 1531681763
    assert oh.seed_from_args(0, 0) == kat["seed_hash"]["0,0"] == 397586535
    assert oh.seed_from_args(0, 1) == kat["seed_hash"]["0,1"] == 16979904
    assert oh.infinite_order(10, 0, 1, 0, 0, 10) == kat["shuffle10_seed0_epoch0"] == [6, 8, 9, 7, 2, 3, 0, 1, 5, 4]
    # rank r of R yields order[(start + r + j R) % N]
    full = oh.infinite_order(10, 0, 1, 0, 0, 10)
    assert oh.infinite_order(10, 1, 2, 0, 2, 3) == [full[3], full[5], full[7]]
    for n, ref in zip((0, 250, 999), kat["lr_linear"]):
        assert oh.linear_lr(n, 1000, 1e-4) == pytest.approx(ref, rel=1e-12)


def test_layernorm(golden_dir):
    g = _load(golden_dir, "ops.npz")
    assert torch.allclose(ou.channel_layer_norm(_t(g["ln4_x"]), 1), _t(g["ln4_y"]), atol=1e-6)
    assert torch.allclose(ou.channel_layer_norm(_t(g["ln3_x"]), 1), _t(g["ln3_y"]), atol=1e-6)
    # the switch exists and matters (biased differs by sqrt((C-1)/C))
    yb = ou.channel_layer_norm(_t(g["ln4_x"]), 1, ln_unbiased=False)
    assert not torch.allclose(yb, _t(g["ln4_y"]), atol=1e-3)


def test_res_block_fwd_bwd(golden_dir):
    g = _load(golden_dir, "ops.npz")
    sd = {"b." + k[len("res_p."):]: _t(v).requires_grad_(True) for k, v in g.items() if k.startswith("res_p.")}
    x = _t(g["res_x"]).requires_grad_(True)
    e = _t(g["res_e"]).requires_grad_(True)
    y = ou.mod_res_block(sd, "b.", x, e, F.silu)
    assert torch.allclose(y, _t(g["res_y"]), atol=2e-6)
    names = list(sd)
    grads = torch.autograd.grad((y * _t(g["res_w"])).sum(), [x, e] + [sd[n] for n in names])
    assert torch.allclose(grads[0], _t(g["res_gx"]), atol=1e-5)
    assert torch.allclose(grads[1], _t(g["res_ge"]), atol=1e-4)
    for n, gr in zip(names, grads[2:]):
        ref = _t(g["res_g." + n[2:]])
        assert torch.allclose(gr, ref, atol=1e-4 * max(1.0, ref.abs().max().item())), n


def test_attention_block_fwd_bwd(golden_dir):
    g = _load(golden_dir, "ops.npz")
    sd = {"a." + k[len("att_p."):]: _t(v).requires_grad_(True) for k, v in g.items() if k.startswith("att_p.")}
    x = _t(g["att_x"]).requires_grad_(True)
    y = ou.attention_block(sd, "a.", x)
    assert torch.allclose(y, _t(g["att_y"]), atol=2e-6)
    names = list(sd)
    grads = torch.autograd.grad((y * _t(g["att_w"])).sum(), [x] + [sd[n] for n in names])
    assert torch.allclose(grads[0], _t(g["att_gx"]), atol=1e-5)
    for n, gr in zip(names, grads[1:]):
        assert torch.allclose(gr, _t(g["att_g." + n[2:]]), atol=1e-4), n


def _tiny_sd(g):
    return {k[3:]: _t(v) for k, v in g.items() if k.startswith("sd.")}


def test_tiny_net_forward_loss_grads(golden_dir):
    g = _load(golden_dir, "tiny_net.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in _tiny_sd(g).items()}
    x, t, eps = _t(g["x"]), _t(g["t"]), _t(g["eps"])
    xt = od.perturb(x, t, eps)
    assert torch.allclose(xt, _t(g["xt"]), atol=1e-6)
    net = lambda a, b: ou.score_unet_forward(sd, a, b, **TINY)
    y = net(xt, t)
    assert torch.allclose(y, _t(g["y"]), atol=1e-5)
    loss = od.loss(net, x, t, eps).mean()
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-5)
    names = [str(n) for n in g["param_order"]]
    grads = torch.autograd.grad(loss, [sd[n] for n in names])
    for n, gr in zip(names, grads):
        ref = _t(g["grad." + n])
        assert torch.allclose(gr, ref, atol=1e-5 + 1e-4 * ref.abs().max().item()), n
    with torch.no_grad():
        y32 = ou.score_unet_forward(sd, _t(g["x32"]), torch.tensor(0.3), **TINY)
    assert torch.allclose(y32, _t(g["y32"]), atol=1e-5)


def test_forcing_projection_against_the_reference(golden_dir):
    """model/score.py:49-51,65-66 (forcing_dim > 0): output, loss and gradients of the imported reference on the tiny network with a
    5-wide forcing vector; the weights are rebuilt from the seed (map_forcing is created first: creation-order parity)."""
    from climate2weather_amd.score import ScoreUNet
    g = _load(golden_dir, "tiny_net_forcing.npz")
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, forcing_dim=5, **{**TINY_CFG})
    assert list(net.state_dict().keys())[:2] == ["map_forcing.weight", "map_forcing.bias"]
    assert sum(v.double().abs().sum().item() for v in net.state_dict().values()) == pytest.approx(float(g["sd_abs_sum"]), rel=1e-12)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    x, t, eps, forcing = (_t(g[k]) for k in ("x", "t", "eps", "forcing"))
    y = ou.score_unet_forward(sd, x, t, forcing=forcing, **TINY)
    assert torch.allclose(y, _t(g["y"]), atol=1e-5)
    loss = ((y - eps) ** 2).mean()
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-5)
    names = [str(n) for n in g["names"]]
    grads = dict(zip(names, torch.autograd.grad(loss, [sd[n] for n in names])))
    for n, ref in zip(names, g["grad_norm"]):
        assert grads[n].double().norm().item() == pytest.approx(float(ref), rel=2e-4, abs=1e-9), n
    for n in ("map_forcing.weight", "map_forcing.bias", "map_layer1.weight", "map_layer0.bias", "unet.heads.0.weight"):
        ref = _t(g["grad." + n])
        assert torch.allclose(grads[n], ref, atol=1e-6 + 1e-4 * ref.abs().max().item()), n
    with torch.no_grad():
        ys = ou.score_unet_forward(sd, x[:1], torch.tensor(0.3), forcing=forcing[:1], **TINY)
    assert torch.allclose(ys, _t(g["y_scalar_t"]), atol=1e-5)


def test_window_fold_unfold_and_sampler(golden_dir):
    g = _load(golden_dir, "tiny_net.npz")
    s = _load(golden_dir, "sampler.npz")
    sd = _tiny_sd(g)
    net = lambda a, b: ou.score_unet_forward(sd, a, b, **TINY)
    k = 1
    with torch.no_grad():
        y = od.window_score(net, _t(s["score_x"]), torch.tensor(0.7), k)
        yb = od.window_score(net, _t(s["score_x"]), torch.tensor(0.7), k, batch_size=4)
    assert torch.allclose(y, _t(s["score_y"]), atol=1e-5)
    assert torch.allclose(yb, _t(s["score_y"]), atol=1e-5)

    def A(x):
        return F.avg_pool2d(x[::2], 8)

    y_obs, std, gamma = _t(s["y_obs"]), _t(s["std"]), float(s["gamma"])
    for name, corrections, cond, exact in [("uncond_c0", 0, False, False), ("uncond_c1", 1, False, False),
                                           ("cond_c0", 0, True, False), ("cond_c1_exact", 1, True, True)]:
        fn = od.GuidedScore(net, k, A if cond else None, y_obs, std, gamma, exact, batch_size=4)
        zs = [_t(z) for z in s[name + ".z"]] if corrections else None
        x = od.sample(fn, _t(s[name + ".noise"]), steps=4, corrections=corrections, tau=0.5, z_draws=zs)
        ref = _t(s[name + ".x"])
        assert (x - ref).abs().max().item() <= 2e-4 * ref.abs().max().item(), name


def test_sampler_with_per_variable_gamma(golden_dir):
    """gamma as the (1, F, 1, 1) tensor exp/downscaling.py:228-233 builds for a list-valued likelihood_gamma
    (src/thor/score.py:55 broadcasts it): trajectories of the imported reference, tests/golden/make_golden.py::sampler_tensor_gamma."""
    g = _load(golden_dir, "tiny_net.npz")
    s = _load(golden_dir, "sampler.npz")
    sg = _load(golden_dir, "sampler_gamma.npz")
    sd = _tiny_sd(g)
    net = lambda a, b: ou.score_unet_forward(sd, a, b, **TINY)

    def A(x):
        return F.avg_pool2d(x[::2], 8)

    y_obs, std, gamma = _t(s["y_obs"]), _t(s["std"]), _t(sg["gamma"])
    assert gamma.shape == (1, 2, 1, 1)
    for name, corrections, exact in [("cond_c0_gvec", 0, False), ("cond_c1_gvec_exact", 1, True)]:
        fn = od.GuidedScore(net, 1, A, y_obs, std, gamma, exact, batch_size=4)
        zs = [_t(z) for z in sg[name + ".z"]] if corrections else None
        x = od.sample(fn, _t(s["cond_c0.noise"]), steps=4, corrections=corrections, tau=0.5, z_draws=zs)
        ref = _t(sg[name + ".x"])
        assert (x - ref).abs().max().item() <= 2e-4 * ref.abs().max().item(), name
    fn = od.GuidedScore(net, 1, A, y_obs, std, gamma, False, batch_size=4)
    out = fn(_t(s["score_x"]), torch.tensor(0.7))
    ref = _t(sg["score_guided_gvec"])
    assert (out - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


def test_ema_and_adamw(golden_dir):
    g = _load(golden_dir, "ema.npz")
    for rate in (0.9, 0.999):
        assert torch.allclose(oh.ema_update(_t(g["p0"]), _t(g["p1"]), rate), _t(g[f"ema_{rate}"]), atol=1e-6)
    # AdamW restatement against torch.optim.AdamW (train.py:176-181 hyper-parameters)
    torch.manual_seed(0)
    p = torch.nn.Parameter(torch.randn(50))
    opt = torch.optim.AdamW([p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3)
    q, m, v = p.detach().clone(), torch.zeros(50), torch.zeros(50)
    for step in range(1, 4):
        gr = torch.randn(50)
        p.grad = gr.clone()
        opt.step()
        q, m, v = oh.adamw_step(q, gr, m, v, step, 1e-3)
        assert torch.allclose(q, p.detach(), atol=1e-6)
    assert oh.batch_split(512, 4, 128) == (128, 1)
    assert oh.batch_split(512, 2, 128) == (128, 2)


@pytest.mark.parametrize("C", [52, 65])
def test_full_size_backward_fingerprints(golden_dir, C):
    """The oracle itself against the full-size gradient fixtures of the imported reference (make_golden.py::full_net_gradients):
    default network, B = 2, injected (t, eps) -- loss, L2 norm / absolute sum of all 228 gradients, the kept slices."""
    from climate2weather_amd.score import ScoreUNet  # parameter container only: creation-order init == the reference's (tested)
    g = {k: v for k, v in np.load(os.path.join(golden_dir, f"full_net_grads_c{C}.npz"), allow_pickle=False).items()}
    cfg = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3, padding_mode="zeros",
               attention_levels=[4])
    torch.manual_seed(0)
    net = ScoreUNet(channels=C, spatial=2, activation=torch.nn.SiLU, **cfg)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    gen = torch.Generator().manual_seed(int(g["seed"]))
    x = torch.randn(2, C, 128, 128, generator=gen) * 0.5 + 0.5
    t = torch.rand(2, 1, 1, 1, generator=gen)
    eps = torch.randn(2, C, 128, 128, generator=gen)
    assert eps.double().sum().item() == pytest.approx(float(g["eps_checksum"]), rel=1e-12)
    fwd = lambda a, b: ou.score_unet_forward(sd, a, b, cfg["hidden_blocks"], cfg["attention_levels"])
    loss = od.loss(fwd, x, t, eps).mean()
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-6)
    names = [str(n) for n in g["names"]]
    grads = dict(zip(sd.keys(), torch.autograd.grad(loss, list(sd.values()))))
    for n, nr, ab in zip(names, g["norm"], g["abs_sum"]):
        assert grads[n].double().norm().item() == pytest.approx(float(nr), rel=1e-4), n
        assert grads[n].double().abs().sum().item() == pytest.approx(float(ab), rel=1e-4), n
    for k in g:
        if k.startswith("slice."):
            n = k[len("slice."):]
            s0, s1 = (int(v) for v in g["step." + n])
            ref = torch.from_numpy(g[k])
            assert (grads[n][::s0, ::s1] - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-12, n


def test_tiny_net_with_the_unets_default_relu(golden_dir):
    """model/nn.py:118: the UNet's own default activation is ReLU (train.py:171 passes SiLU).  Oracle (act=F.relu) vs the imported
    reference built with activation=torch.nn.ReLU on the same weights and inputs."""
    g = {k: v for k, v in np.load(os.path.join(golden_dir, "tiny_net.npz"), allow_pickle=False).items()}
    r = {k: v for k, v in np.load(os.path.join(golden_dir, "tiny_net_relu.npz"), allow_pickle=False).items()}
    sd = {k[3:]: torch.from_numpy(v).clone().requires_grad_(True) for k, v in g.items() if k.startswith("sd.")}
    x, t, eps = (torch.from_numpy(g[k]) for k in ("x", "t", "eps"))
    fwd = lambda a, b: ou.score_unet_forward(sd, a, b, hidden_blocks=[1, 1], attention_levels=[1], act=F.relu)
    y = fwd(od.perturb(x, t, eps), t)
    assert torch.allclose(y, torch.from_numpy(r["y"]), atol=2e-5)
    loss = ((y - eps) ** 2).mean()
    assert loss.item() == pytest.approx(float(r["loss"]), rel=1e-5)
    grads = torch.autograd.grad(loss, list(sd.values()))
    for (k, _), gr in zip(sd.items(), grads):
        ref = torch.from_numpy(r["grad." + k])
        assert torch.allclose(gr, ref, atol=1e-6 + 2e-4 * ref.abs().max().item()), k
