"""Shared body of the CPU-emulated and the GPU test that resume from a reference-made training-state checkpoint."""
import os

import numpy as np
import pytest
import torch

from climate2weather_amd.training import Trainer


def resume_from_reference_checkpoint(golden_dir, dev, make_net):
    """A `training-state-*.ckpt` with the reference's contents (tests/golden/make_checkpoint.py: the imported reference's module with
    zuko's persistent `eps` buffers, torch AdamW, the reference's StandardEMA, after ONE step) -> Trainer.load_state_dict -> one more
    step == the reference's second step (src/thor/checkpoint.py:37-57, training_loop.py:132-139)."""
    from climate2weather_amd.ema import StandardEMA as EMA
    g = {k: v for k, v in np.load(os.path.join(golden_dir, "tiny_net.npz"), allow_pickle=False).items()}
    nxt = {k: v for k, v in np.load(os.path.join(golden_dir, "ref_training_state_tiny_next.npz"), allow_pickle=False).items()}
    ck = torch.load(os.path.join(golden_dir, "ref_training_state_tiny.ckpt"), map_location=dev, weights_only=True)
    n_eps = sum(k.endswith(".eps") for k in ck["net"])
    assert n_eps == int(nxt["n_eps_keys"]) == 7 and len(ck["net"]) == 7 + len(list(make_net(0).parameters()))
    net = make_net(99)
    tr = Trainer(net, lr=1e-3, precision="fp32", ema_rates=[0.9])
    tr.load_state_dict(ck)
    assert tr.cur_ndata == 2 and tr.step_count == 1 and tr.total_elapsed_time == 1.5
    for k, v in net.state_dict().items():
        assert torch.equal(v, ck["net"][k]), k
    x, t, eps = (torch.from_numpy(g[k]).to(dev) for k in ("x", "t", "eps"))
    loss = tr.step(x, t=t.reshape(-1), eps=eps)
    assert float(loss) == pytest.approx(float(nxt["loss2"]), rel=2e-4)
    # Adam's update is lr * m_hat / (sqrt(v_hat) + eps): where the gradient is ~eps it is ill-conditioned; bound those by 2 lr
    gr = {k: (ck["net"][k].cpu() - torch.from_numpy(nxt["p." + k])).abs() for k in net.state_dict()}
    for k, v in net.state_dict().items():
        ref = torch.from_numpy(nxt["p." + k])
        moved = gr[k] > 2e-4  # entries whose reference update is a full-size Adam step
        assert torch.allclose(v.cpu()[moved], ref[moved], atol=2e-5), k
        assert (v.cpu() - ref).abs().max().item() <= 2e-3, k
    for rate, sd in tr.ema_state_dicts():
        for k, v in sd.items():
            assert (v.cpu() - torch.from_numpy(nxt["ema." + k])).abs().max().item() <= 2.1e-4, k  # 0.1 x the bound above
    # the module API takes the same file: strict load with the `*.eps` keys present, StandardEMA.load_state_dict, torch AdamW
    net2 = make_net(5)
    net2.load_state_dict(ck["net"])
    ema2 = EMA(net2, rates=[0.5])
    ema2.load_state_dict(ck["ema"])
    assert ema2.rates == [0.9]
    for k, v in ema2.emas[0].state_dict().items():
        assert torch.equal(v, ck["ema"]["emas"][0][k]), k
    torch.optim.AdamW(net2.parameters(), lr=1e-3).load_state_dict(ck["optimizer"])
    bad = dict(ck["net"])
    bad["unet.tails.0.0.eps"] = torch.tensor(1e-3)
    with pytest.raises(ValueError, match="eps"):
        net2.load_state_dict(bad)
    bad = dict(ck["net"])
    bad["unet.tails.0.0.gamma"] = torch.tensor(1.0)
    with pytest.raises(RuntimeError):  # anything else unexpected stays an error
        net2.load_state_dict(bad)
