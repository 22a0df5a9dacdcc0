"""The reference's step loop in miniature (training_loop.py:369-391), shared by the CPU-emulated and the GPU tests of the drop-in
optimizer / fused module-path loss."""
import contextlib

import torch


def _batch(i, dev="cpu", shape=(2, 6, 16, 16)):
    g = torch.Generator().manual_seed(100 + i)
    return (torch.randn(*shape, generator=g) * 0.5 + 0.5).to(dev)


def _close(a, b, tol=1e-4):
    """scale-relative: max |a - b| <= tol * max |b| (moments have entries around zero)"""
    return (a - b).abs().max().item() <= tol * b.abs().max().item() + 1e-30


def _loop(net, opt, pipe, steps, lr_fn=None, scaler=None, mirror=None, first=0, dev="cpu", shape=(2, 6, 16, 16), autocast=None):
    """training_loop.py:369-391 in miniature: zero_grad -> loss.mean().backward() -> lr into the groups -> step.
    ``mirror`` = (net_b, opt_b): a second optimizer stepped on THE SAME gradients (copied over, unscaled), never on its own backward --
    Adam turns the round-off of a vanishing gradient (the key bias of an attention block) into a full-size step, so two loops that
    each differentiate for themselves part ways on such entries whatever the optimizer."""
    losses = []
    for i in range(first, first + steps):
        opt.zero_grad()
        torch.manual_seed(1000 + i)
        with (torch.autocast(torch.device(dev).type, dtype=autocast) if autocast is not None else contextlib.nullcontext()):
            loss = pipe.loss(net=net, x=_batch(i, dev, shape)).mean().mul(1.0)
        (scaler.scale(loss) if scaler is not None else loss).backward()
        if mirror is not None:
            inv = 1.0 / scaler.get_scale() if scaler is not None else 1.0
            for p, q in zip(net.parameters(), mirror[0].parameters()):
                q.grad = p.grad.detach().clone() * inv
        for o in (opt,) + ((mirror[1],) if mirror is not None else ()):
            if lr_fn is not None:
                for g in o.param_groups:
                    g["lr"] = lr_fn(i)
        if scaler is not None:
            scaler.step(opt)
            scaler.update()
        else:
            opt.step()
        if mirror is not None:
            mirror[1].step()
        losses.append(loss.detach().item())
    return losses


