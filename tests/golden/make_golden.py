#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference, read-only).  The
reference's Python never travels: what is committed are the inputs and expected
outputs below, plus this script.  `zuko` (un-vendored third party) is replaced by
oracle/_shim/zuko -- see that file's header for the "parity unpinned" caveat.

    python tests/golden/make_golden.py
"""
import importlib.util
import json
import math
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [os.path.join(REPO, "oracle", "_shim"), REF]

from model.score import ScoreUNet, timestep_embedding  # noqa: E402  (reference)
from model.nn import AttentionBlock, ModResidualBlock, UNet  # noqa: E402  (reference)
from zuko.nn import LayerNorm  # noqa: E402  (shim)


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


pipelines = load_by_path("ref_pipelines", f"{REF}/src/thor/pipelines.py")
score = load_by_path("ref_score", f"{REF}/src/thor/score.py")
ema_mod = load_by_path("ref_ema", f"{REF}/src/thor/ema.py")
lr_mod = load_by_path("ref_lr", f"{REF}/src/thor/lr.py")

torch.set_num_threads(8)
CPU = torch.device("cpu")

TINY = dict(embedding_dim=64, hidden_channels=[32, 64], hidden_blocks=[1, 1], attention_levels=[1],
            kernel_size=3, padding_mode="zeros")


def npd(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def kats():
    out = {}
    t = torch.tensor([0.0, 0.25, 0.5, 1.0])
    out["timestep_embedding_t"] = t.tolist()
    out["timestep_embedding"] = timestep_embedding(t, 32).tolist()
    p = pipelines.SDAPipeline()
    out["eta"] = p.eta
    out["mu"] = p.mu(t).tolist()
    out["sigma"] = p.sigma(t).tolist()
    out["lr_linear"] = [lr_mod.linear_learning_rate_schedule(n, 1000, 1e-4) for n in (0, 250, 999)]
    # util.py:27-29 / dataset.py:34-38 quoted expressions (files not importable: lightning/h5py missing)
    out["seed_hash"] = {"42,0": hash((42, 0)) % (1 << 31), "0,0": hash((0, 0)) % (1 << 31), "0,1": hash((0, 1)) % (1 << 31)}
    order = np.arange(10)
    np.random.RandomState(hash((0, 0)) % (1 << 31)).shuffle(order)
    out["shuffle10_seed0_epoch0"] = order.tolist()
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(out, f, indent=1)


def ops():
    """Per-op vectors: channel LN (both call shapes), ModResidualBlock, AttentionBlock, with input/param grads."""
    g = torch.Generator().manual_seed(7)
    out = {}
    x = torch.randn(2, 32, 8, 8, generator=g)
    out["ln4_x"] = x.numpy()
    out["ln4_y"] = LayerNorm(-3)(x).numpy()
    x3 = torch.randn(2, 32, 16, generator=g)
    out["ln3_x"] = x3.numpy()
    out["ln3_y"] = LayerNorm(1)(x3).numpy()

    torch.manual_seed(11)
    unet = UNet(6, 6, 64, hidden_channels=[32], hidden_blocks=[1], activation=torch.nn.SiLU, spatial=2,
                kernel_size=3, padding_mode="zeros")
    blk = unet.descent[0][0]
    assert isinstance(blk, ModResidualBlock)
    x = torch.randn(2, 32, 8, 8, generator=g, requires_grad=True)
    e = torch.randn(2, 64, generator=g, requires_grad=True)
    y = blk(x, e)
    w = torch.randn(y.shape, generator=g)
    grads = torch.autograd.grad((y * w).sum(), [x, e] + list(blk.parameters()))
    out.update({"res_x": x.detach().numpy(), "res_e": e.detach().numpy(), "res_y": y.detach().numpy(), "res_w": w.numpy(),
                "res_gx": grads[0].numpy(), "res_ge": grads[1].numpy()})
    for (k, v), gr in zip(blk.named_parameters(), grads[2:]):
        out["res_p." + k] = v.detach().numpy()
        out["res_g." + k] = gr.numpy()

    torch.manual_seed(12)
    att = AttentionBlock(32)
    x = torch.randn(2, 32, 4, 4, generator=g, requires_grad=True)
    y = att(x, None)
    w = torch.randn(y.shape, generator=g)
    grads = torch.autograd.grad((y * w).sum(), [x] + list(att.parameters()))
    out.update({"att_x": x.detach().numpy(), "att_y": y.detach().numpy(), "att_w": w.numpy(), "att_gx": grads[0].numpy()})
    for (k, v), gr in zip(att.named_parameters(), grads[1:]):
        out["att_p." + k] = v.detach().numpy()
        out["att_g." + k] = gr.numpy()
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **out)


def tiny_net():
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **TINY)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 6, 16, 16, generator=g) * 0.5 + 0.5
    t = torch.rand(2, 1, 1, 1, generator=g)
    eps = torch.randn(2, 6, 16, 16, generator=g)
    pipe = pipelines.SDAPipeline()
    xt = pipe.mu(t) * x + pipe.sigma(t) * eps
    y = net(xt, t)
    loss = ((y - eps) ** 2).mean()
    names = [k for k, _ in net.named_parameters()]
    grads = torch.autograd.grad(loss, list(net.parameters()))
    out = {"x": x.numpy(), "t": t.numpy(), "eps": eps.numpy(), "xt": xt.detach().numpy(), "y": y.detach().numpy(),
           "loss": np.array(loss.item(), dtype=np.float64)}
    for k, v in net.state_dict().items():
        out["sd." + k] = v.numpy()
    for k, gr in zip(names, grads):
        out["grad." + k] = gr.numpy()
    out["param_order"] = np.array(names)
    # scalar-t call (sampling path: t is 0-d, model/score.py:60)
    x32 = torch.randn(3, 6, 32, 32, generator=g)
    with torch.no_grad():
        y32 = net(x32, torch.tensor(0.3))
    out["x32"] = x32.numpy()
    out["y32"] = y32.numpy()
    np.savez_compressed(os.path.join(HERE, "tiny_net.npz"), **out)
    return net


def tiny_net_relu():
    """The same tiny network (same seed, hence the weights of tiny_net.npz) with the reference UNet's OWN default activation,
    torch.nn.ReLU (model/nn.py:118; train.py:171 passes SiLU): output, loss and every gradient on tiny_net.npz's inputs."""
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.ReLU, **TINY)
    g0 = np.load(os.path.join(HERE, "tiny_net.npz"))
    for k, v in net.state_dict().items():
        assert np.array_equal(v.numpy(), g0["sd." + k]), k
    x, t, eps = (torch.from_numpy(g0[k]) for k in ("x", "t", "eps"))
    pipe = pipelines.SDAPipeline()
    xt = pipe.mu(t) * x + pipe.sigma(t) * eps
    y = net(xt, t)
    loss = ((y - eps) ** 2).mean()
    names = [k for k, _ in net.named_parameters()]
    grads = torch.autograd.grad(loss, list(net.parameters()))
    out = {"y": y.detach().numpy(), "loss": np.array(loss.item(), dtype=np.float64)}
    for k, gr in zip(names, grads):
        out["grad." + k] = gr.numpy()
    with torch.no_grad():
        out["y32"] = net(torch.from_numpy(g0["x32"]), torch.tensor(0.3)).numpy()
    np.savez_compressed(os.path.join(HERE, "tiny_net_relu.npz"), **out)


def tiny_net_forcing():
    """model/score.py:49-51,65-66: the forcing projection (no caller in the reference passes forcing, the code path exists): the tiny
    network with forcing_dim = 5 -- map_forcing is created FIRST, so the same seed gives other weights than tiny_net.npz; the tests
    rebuild them from the seed (creation-order parity) and only fingerprints travel -- output, loss, the L2 norm of every gradient,
    and the gradients of map_forcing / map_layer1 / one conv in full; plus a scalar-t call with one forcing row."""
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, forcing_dim=5, **TINY)
    g = torch.Generator().manual_seed(55)
    x = torch.randn(2, 6, 16, 16, generator=g) * 0.5 + 0.5
    t = torch.rand(2, generator=g)
    eps = torch.randn(2, 6, 16, 16, generator=g)
    forcing = torch.randn(2, 5, generator=g)
    y = net(x, t, forcing=forcing)
    loss = ((y - eps) ** 2).mean()
    names = [k for k, _ in net.named_parameters()]
    grads = dict(zip(names, torch.autograd.grad(loss, list(net.parameters()))))
    out = {"x": x.numpy(), "t": t.numpy(), "eps": eps.numpy(), "forcing": forcing.numpy(), "y": y.detach().numpy(),
           "loss": np.array(loss.item(), dtype=np.float64), "names": np.array(names),
           "sd_abs_sum": np.array(sum(v.double().abs().sum().item() for v in net.state_dict().values())),
           "grad_norm": np.array([grads[k].double().norm().item() for k in names])}
    for k in ("map_forcing.weight", "map_forcing.bias", "map_layer1.weight", "map_layer0.bias", "unet.heads.0.weight"):
        out["grad." + k] = grads[k].numpy()
    with torch.no_grad():
        out["y_scalar_t"] = net(x[:1], torch.tensor(0.3), forcing=forcing[:1]).numpy()
    np.savez_compressed(os.path.join(HERE, "tiny_net_forcing.npz"), **out)
    print("tiny_net_forcing: loss", loss.item(), "keys", list(net.state_dict().keys())[:2])


def sampler(net):
    """Sampler trajectories on the tiny net: L=9, F=2, k=1, 32^2, 4 steps."""
    pipe = pipelines.SDAPipeline()
    out = {}
    L, Fv, k, H = 9, 2, 1, 32

    def A(x):
        return torch.nn.functional.avg_pool2d(x[::2], 8)

    torch.manual_seed(21)
    truth = torch.randn(L, Fv, H, H) * 0.3 + 0.5
    y_obs = A(truth)
    std = torch.tensor([0.8, 0.6]).reshape(1, Fv, 1, 1)  # mild guidance: an untrained net + tight std blows up
    gamma = 1e-2
    out.update({"y_obs": y_obs.numpy(), "std": std.numpy(), "gamma": np.array(gamma)})

    for name, corrections, cond, exact in [("uncond_c0", 0, False, False), ("uncond_c1", 1, False, False),
                                           ("cond_c0", 0, True, False), ("cond_c1_exact", 1, True, True)]:
        torch.manual_seed(1)
        noise = torch.randn(L, Fv, H, H)
        state = torch.get_rng_state()
        sf = score.DefaultScoreFunction(net, markov_order=k, noise_process=pipe)
        sfb = score.BatchedScoreFunction(net, markov_order=k, batch_size=4, device=CPU, noise_process=pipe)
        if cond:
            sf.condition_on(A=A, y=y_obs, std=std, gamma=gamma, exact_grad=exact)
            sfb.condition_on(A=A, y=y_obs, std=std, gamma=gamma, exact_grad=exact)
        xs = pipe.sample(sf, noise, steps=4, corrections=corrections, tau=0.5, device=CPU, show_progressbar=False)
        torch.set_rng_state(state)
        xb = pipe.sample(sfb, noise, steps=4, corrections=corrections, tau=0.5, device=CPU, show_progressbar=False)
        assert torch.allclose(xs, xb, rtol=1e-4, atol=1e-4), (name, (xs - xb).abs().max())
        print(name, "max|x|", xs.abs().max().item())
        torch.set_rng_state(state)
        zs = [torch.empty_like(noise).normal_().numpy() for _ in range(4 * corrections)]
        out[name + ".noise"] = noise.numpy()
        out[name + ".x"] = xs.numpy()
        if zs:
            out[name + ".z"] = np.stack(zs)
    # one raw score evaluation (unfold -> net -> fold) for window plumbing
    with torch.no_grad():
        sf = score.DefaultScoreFunction(net, markov_order=k, noise_process=pipe)
        out["score_x"] = noise.numpy()
        out["score_y"] = sf(noise, torch.tensor(0.7)).numpy()
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **out)


def full_net_fingerprint():
    """Default config (configs/sda_unet.yml), C=52: too large to ship -> fingerprints."""
    cfg = yaml.full_load(open(f"{REF}/configs/sda_unet.yml"))
    torch.manual_seed(0)
    net = ScoreUNet(channels=52, spatial=2, activation=torch.nn.SiLU, **cfg)
    sd = net.state_dict()
    fp = {"n_params": sum(p.numel() for p in net.parameters()), "n_tensors": len(sd),
          "heads0_sum": sd["unet.heads.0.weight"].double().sum().item(),
          "abs_sum": sum(v.double().abs().sum().item() for v in sd.values()),
          "keys": list(sd.keys()), "shapes": [list(v.shape) for v in sd.values()]}
    x = torch.randn(1, 52, 128, 128, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        y = net(x, torch.tensor([0.3]))
    fp["y_mean"] = y.double().mean().item()
    fp["y_std"] = y.double().std().item()
    fp["y_000"] = y[0, 0, 0, :3].tolist()
    with open(os.path.join(HERE, "full_net_fingerprint.json"), "w") as f:
        json.dump(fp, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "full_net_slice.npz"), y_slice=y[:, :, ::16, ::16].numpy())


def sampler_tensor_gamma():
    """Guided trajectories with the per-variable ``gamma`` the reference driver builds for a list-valued ``likelihood_gamma``
    (exp/downscaling.py:228-233: ``torch.zeros(1, C, 1, 1)`` filled per variable; src/thor/score.py:55 broadcasts it): the same tiny
    net, observation, std and noise draws as sampler(), written to a file of its own so sampler.npz stays byte for byte what it was."""
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **TINY)
    g0 = np.load(os.path.join(HERE, "tiny_net.npz"))
    for k, v in net.state_dict().items():
        assert np.array_equal(v.numpy(), g0["sd." + k]), k
    s0 = np.load(os.path.join(HERE, "sampler.npz"))
    pipe = pipelines.SDAPipeline()
    L, Fv, k, H = 9, 2, 1, 32

    def A(x):
        return torch.nn.functional.avg_pool2d(x[::2], 8)

    y_obs, std = torch.from_numpy(s0["y_obs"]), torch.from_numpy(s0["std"])
    gamma = torch.zeros(1, Fv, 1, 1)
    for c, gv in enumerate([1e-2, 2.5e-1]):  # as exp/downscaling.py:229-231 fills it
        gamma[:, c, ...] = gv
    out = {"gamma": gamma.numpy()}
    for name, corrections, exact in [("cond_c0_gvec", 0, False), ("cond_c1_gvec_exact", 1, True)]:
        torch.manual_seed(1)
        noise = torch.randn(L, Fv, H, H)
        assert np.array_equal(noise.numpy(), s0["cond_c0.noise"])
        state = torch.get_rng_state()
        sf = score.DefaultScoreFunction(net, markov_order=k, noise_process=pipe)
        sfb = score.BatchedScoreFunction(net, markov_order=k, batch_size=4, device=CPU, noise_process=pipe)
        sf.condition_on(A=A, y=y_obs, std=std, gamma=gamma, exact_grad=exact)
        sfb.condition_on(A=A, y=y_obs, std=std, gamma=gamma, exact_grad=exact)
        xs = pipe.sample(sf, noise, steps=4, corrections=corrections, tau=0.5, device=CPU, show_progressbar=False)
        torch.set_rng_state(state)
        xb = pipe.sample(sfb, noise, steps=4, corrections=corrections, tau=0.5, device=CPU, show_progressbar=False)
        assert (xs - xb).abs().max() <= 1e-5 * xs.abs().max(), (name, (xs - xb).abs().max())  # Default == Batched up to fp32 summation order
        torch.set_rng_state(state)
        zs = [torch.empty_like(noise).normal_().numpy() for _ in range(4 * corrections)]
        out[name + ".x"] = xs.numpy()
        if zs:
            out[name + ".z"] = np.stack(zs)
        print(name, "max|x|", xs.abs().max().item(), "vs scalar-gamma trajectory", np.abs(xs.numpy() - s0["cond_c0.x"]).max())
    # one guided score evaluation (the term the fused kernel computes), exact_grad=False
    sf = score.DefaultScoreFunction(net, markov_order=k, noise_process=pipe)
    sf.condition_on(A=A, y=y_obs, std=std, gamma=gamma, exact_grad=False)
    out["score_guided_gvec"] = sf(torch.from_numpy(s0["score_x"]), torch.tensor(0.7)).numpy()
    np.savez_compressed(os.path.join(HERE, "sampler_gamma.npz"), **out)


# Representative gradient tensors of the default network kept as strided slices: the network-input conv, a residual
# conv at full resolution, a stride-2 head, the up-conv back to full resolution, the output conv, attention qkv / proj, a modulation
# projection and the time MLP.  (step over dim 0, step over dim 1)
FULL_GRAD_SLICES = {
    "unet.heads.0.weight": (4, 1), "unet.descent.0.1.residue.1.weight": (8, 8), "unet.ascent.4.2.residue.3.weight": (8, 8),
    "unet.heads.1.0.weight": (8, 8), "unet.heads.4.0.weight": (32, 24), "unet.tails.3.2.weight": (8, 8), "unet.tails.0.2.weight": (24, 32),
    "unet.tails.4.weight": (1, 8), "unet.descent.2.1.residue.3.weight": (16, 16), "unet.descent.4.2.residue.1.weight": (32, 32),
    "unet.descent.4.1.qkv.weight": (48, 32), "unet.ascent.0.3.proj_out.weight": (32, 32), "unet.descent.0.0.project.0.weight": (8, 32),
    "unet.ascent.2.1.project.0.weight": (16, 32), "map_layer0.weight": (16, 1), "map_layer1.weight": (32, 32),
}


def full_net_gradients():
    """Backward of the default network pinned to the imported reference (training_loop.py:376-378 over src/thor/pipelines.py:27-35 with
    the draws injected): B = 2 at C = 52 (the reference's recipe) and C = 65 (north-star, the benchmarked channel count: exercises the
    65 -> 128 channel padding of the network-input / output convs).  72 M gradients are too many to ship: per tensor the absolute sum
    and the L2 norm (float64) of ALL 228 gradients, strided slices of the tensors in FULL_GRAD_SLICES and of every modulation projection, every bias gradient in
    full, plus the loss and a strided output slice.  Inputs are regenerated by the tests from the seeds below."""
    cfg = yaml.full_load(open(f"{REF}/configs/sda_unet.yml"))
    pipe = pipelines.SDAPipeline()
    for C in (52, 65):
        torch.manual_seed(0)
        net = ScoreUNet(channels=C, spatial=2, activation=torch.nn.SiLU, **cfg)
        g = torch.Generator().manual_seed(4000 + C)
        x = torch.randn(2, C, 128, 128, generator=g) * 0.5 + 0.5
        t = torch.rand(2, 1, 1, 1, generator=g)
        eps = torch.randn(2, C, 128, 128, generator=g)
        xt = pipe.mu(t) * x + pipe.sigma(t) * eps
        y = net(xt, t)
        loss = ((y - eps) ** 2).mean()
        names = [k for k, _ in net.named_parameters()]
        grads = dict(zip(names, torch.autograd.grad(loss, list(net.parameters()))))
        out = {"seed": np.int64(4000 + C), "t": t.numpy(), "loss": np.float64(loss.item()), "y_slice": y.detach()[:, :, ::16, ::16].numpy(),
               "names": np.array(names),
               "abs_sum": np.array([grads[k].double().abs().sum().item() for k in names]),
               "norm": np.array([grads[k].double().norm().item() for k in names]),
               "x_checksum": np.float64(x.double().sum().item()), "eps_checksum": np.float64(eps.double().sum().item())}
        for k, (s0, s1) in FULL_GRAD_SLICES.items():
            out["slice." + k] = grads[k][::s0, ::s1].contiguous().numpy()
            out["step." + k] = np.array([s0, s1])
        # round 6 (parity instrumentation of the 16-bit modes: relative L2 and cosine per tensor need the tensors, not their norms):
        # EVERY bias gradient in full (114 small tensors, 31 k values) and a (4, 8)-strided slice of every modulation projection's
        # weight gradient (30 tensors; these and the biases are sums over all pixels of small per-pixel terms -- the tensors where
        # an error concentrated in small-magnitude entries would show)
        for k in names:
            if k.endswith(".bias"):
                out["full." + k] = grads[k].numpy()
            elif k.endswith("project.0.weight") and k not in FULL_GRAD_SLICES:
                out["slice." + k] = grads[k][::4, ::8].contiguous().numpy()
                out["step." + k] = np.array([4, 8])
        np.savez_compressed(os.path.join(HERE, f"full_net_grads_c{C}.npz"), **out)
        print(f"C={C}: loss {loss.item():.6f}")


def ema_kat():
    torch.manual_seed(4)
    lin = torch.nn.Linear(4, 3)
    ema = ema_mod.StandardEMA(lin, rates=[0.9, 0.999])
    with torch.no_grad():
        for p in lin.parameters():
            p.add_(1.0)
    ema.update()
    out = {"p0": lin.weight.detach().numpy() - 1.0, "p1": lin.weight.detach().numpy(),
           "ema_0.9": ema.emas[0].weight.detach().numpy(), "ema_0.999": ema.emas[1].weight.detach().numpy()}
    np.savez_compressed(os.path.join(HERE, "ema.npz"), **out)


if __name__ == "__main__":
    if sys.argv[1:] == ["full_net_gradients"]:
        full_net_gradients()
    elif sys.argv[1:] == ["tiny_net_relu"]:
        tiny_net_relu()
    elif sys.argv[1:] == ["sampler_tensor_gamma"]:
        sampler_tensor_gamma()
    elif sys.argv[1:] == ["tiny_net_forcing"]:
        tiny_net_forcing()
    else:
        kats()
        ops()
        net = tiny_net()
        tiny_net_relu()
        tiny_net_forcing()
        sampler(net)
        sampler_tensor_gamma()
        ema_kat()
        full_net_fingerprint()
        full_net_gradients()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
