"""Workers for the multi-rank tests of the gradient all-reduce path: "gloo" on the CPU (HIP launchers replaced by tests/emu_ops,
world_size 2) and "nccl" = RCCL on the GPU (the real kernels; world_size 1 on any box -- a real communicator, real collectives
issued from the gradient stream --, world_size 2 where two GPUs are visible)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist


def _init(rank: int, world: int, port: int, backend: str) -> torch.device:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        return dev
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import emu_ops
    from climate2weather_amd import ops as c2w_ops
    for name in emu_ops.ALL:
        if hasattr(c2w_ops, name):
            setattr(c2w_ops, name, getattr(emu_ops, name))
    return torch.device("cpu")


def run(rank: int, world: int, port: int, golden_dir: str, out_dir: str, bucket_mb: float, backend: str = "gloo"):
    dev = _init(rank, world, port, backend)
    from climate2weather_amd.score import ScoreUNet
    from climate2weather_amd.training import Trainer

    g = np.load(os.path.join(golden_dir, "tiny_net.npz"))
    torch.manual_seed(3 + 100 * rank)  # different initial weights per rank: the trainer must broadcast rank 0's
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, embedding_dim=64, hidden_channels=[32, 64], hidden_blocks=[1, 1],
                    attention_levels=[1], kernel_size=3, padding_mode="zeros").to(dev)
    if world == 1:
        os.environ["C2W_FORCE_DIST"] = "1"  # one rank: still broadcast, bucket and all-reduce through the communicator
    tr = Trainer(net, lr=1e-3, precision="fp32", ema_rates=[0.9], bucket_mb=bucket_mb)
    assert tr.sync_grads
    x, t, eps = (torch.from_numpy(g[k]).to(dev) for k in ("x", "t", "eps"))
    per = x.shape[0] // world  # global batch 2 -> one item per rank (both items on a single rank)
    sl = slice(rank * per, (rank + 1) * per)
    loss = tr.step(x[sl].contiguous(), t=t[sl].reshape(-1), eps=eps[sl].contiguous())
    # both ranks seeded identically by their caller: the trainer's own generators must still differ by rank (training_loop.py:49)
    torch.manual_seed(1234)
    same = Trainer(net, lr=0.0, precision="fp32", ema_rates=[], bucket_mb=bucket_mb, seed=7)
    draws = dict(seed=same.rng_cpu.initial_seed(), t=torch.rand(4, generator=same.rng_dev, device=dev).cpu())
    torch.save(dict(loss=float(loss), sd={k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, nb=len(tr.buckets), draws=draws,
                    world=dist.get_world_size(), backend=dist.get_backend()),
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def run_module_ddp(rank: int, world: int, port: int, golden_dir: str, out_dir: str):
    """The reference's own training-loop shape (training_loop.py:116,369-391): the module wrapped in torch DDP, autograd
    backward, torch.optim.AdamW over net.parameters() -- the drop-in seam with nothing replaced but the class name."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import emu_ops
    from climate2weather_amd import ops as c2w_ops
    for name in emu_ops.ALL:
        if hasattr(c2w_ops, name):
            setattr(c2w_ops, name, getattr(emu_ops, name))
    from climate2weather_amd.pipelines import SDAPipeline
    from climate2weather_amd.score import ScoreUNet

    g = np.load(os.path.join(golden_dir, "tiny_net.npz"))
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, embedding_dim=64, hidden_channels=[32, 64], hidden_blocks=[1, 1],
                    attention_levels=[1], kernel_size=3, padding_mode="zeros")
    ddp = torch.nn.parallel.DistributedDataParallel(net)
    opt = torch.optim.AdamW(ddp.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3)
    pipe = SDAPipeline()
    x, t, eps = (torch.from_numpy(g[k]) for k in ("x", "t", "eps"))
    sl = slice(rank, rank + 1)
    opt.zero_grad(set_to_none=True)
    tt = t[sl].reshape(-1, 1, 1, 1)
    xt = pipe.mu(tt) * x[sl] + pipe.sigma(tt) * eps[sl]  # src/thor/pipelines.py:22-25 with the golden noise injected
    loss = ((ddp(xt, t[sl].reshape(-1)) - eps[sl]) ** 2).mean()
    loss.backward()
    grads = {n: p.grad.clone() for n, p in net.named_parameters()}
    opt.step()
    torch.save(dict(loss=float(loss), grads=grads, sd={k: v.clone() for k, v in net.state_dict().items()}),
               os.path.join(out_dir, f"mod{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
