#!/usr/bin/env python3
"""Generate tests/golden/ref_training_state_tiny.ckpt (+ ref_training_state_tiny_next.npz): a `training-state-*.ckpt` with the
contents the reference writes (training_loop.py:132-139,353-363 through src/thor/checkpoint.py:13-35 and `fabric.save`), by
IMPORTING the reference's model / pipeline / EMA classes in the build container and running its training step
(training_loop.py:369-391) on the tiny golden inputs.  The fixture is data (tensors, numbers, key names), not source.

Two things a real reference checkpoint has that the 228-key state_dict of SURVEY.md A1 does not show:
  * zuko registers `eps` as a PERSISTENT buffer, so every LayerNorm contributes a `*.eps` key (40 in the default network, 7 in
    this tiny one) -- the shim's buffer is switched to persistent here before `state_dict()` is taken;
  * `fabric.save` stores `state_dict()` of every stateful object (module, optimizer, StandardEMA) and plain dicts / `__dict__`s
    as they are; Lightning is not installed here, so that container is restated: one `torch.save` of
    {state, net, pipeline, optimizer, ema}.

The companion .npz holds the reference's parameters and EMA after ONE MORE step from the checkpoint (same
injected t / eps), which is what a resumed run must reproduce.

    python tests/golden/make_checkpoint.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [os.path.join(REPO, "oracle", "_shim"), REF]

from model.score import ScoreUNet  # noqa: E402  (reference)
from zuko.nn import LayerNorm  # noqa: E402  (shim)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


pipelines = _load("ref_pipelines_ckpt", f"{REF}/src/thor/pipelines.py")
ema_mod = _load("ref_ema_ckpt", f"{REF}/src/thor/ema.py")

TINY = dict(embedding_dim=64, hidden_channels=[32, 64], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3, padding_mode="zeros")
LR, BATCH = 1e-3, 2


def persistent_eps(net):
    n = 0
    for m in net.modules():
        if isinstance(m, LayerNorm):
            m._non_persistent_buffers_set.discard("eps")
            n += 1
    return n


def step(net, pipe, opt, ema, x, t, eps):
    """training_loop.py:369-391 with the draws of pipeline.loss (src/thor/pipelines.py:27-35) injected."""
    opt.zero_grad(set_to_none=True)
    xt = pipe.mu(t) * x + pipe.sigma(t) * eps
    loss = ((net(xt, t) - eps) ** 2).mean()
    loss.backward()
    for g in opt.param_groups:
        g["lr"] = LR
    opt.step()
    ema.update()
    return float(loss)


def main():
    g = np.load(os.path.join(HERE, "tiny_net.npz"))
    x, t, eps = (torch.from_numpy(g[k]) for k in ("x", "t", "eps"))
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **TINY)
    n_ln = persistent_eps(net)
    opt = torch.optim.AdamW(net.parameters(), lr=LR, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3)  # train.py:176-181
    pipe = pipelines.SDAPipeline()
    ema = ema_mod.StandardEMA(net, rates=[0.9])
    l1 = step(net, pipe, opt, ema, x, t, eps)
    state = dict(cur_ndata=BATCH, total_elapsed_time=1.5)
    ckpt = dict(state=state, net=net.state_dict(), pipeline=pipe.__dict__, optimizer=opt.state_dict(), ema=ema.state_dict())
    assert sum(k.endswith(".eps") for k in ckpt["net"]) == n_ln and len(ckpt["net"]) == len(list(net.parameters())) + n_ln
    path = os.path.join(HERE, "ref_training_state_tiny.ckpt")
    torch.save(ckpt, path)
    l2 = step(net, pipe, opt, ema, x, t, eps)
    out = {"loss1": np.float32(l1), "loss2": np.float32(l2), "n_eps_keys": np.int64(n_ln)}
    for i, (k, p) in enumerate(net.named_parameters()):
        out["p." + k] = p.detach().numpy()  # depends on the restored AdamW moments and step count: they are checked through it
    for k, p in ema.emas[0].named_parameters():
        out["ema." + k] = p.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "ref_training_state_tiny_next.npz"), **out)
    print("wrote", os.path.getsize(path), "bytes;", n_ln, "eps keys; losses", l1, l2)


if __name__ == "__main__":
    main()
