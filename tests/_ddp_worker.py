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


def run_chase(rank: int, world: int, port: int, golden_dir: str, out_dir: str, backend: str = "gloo"):
    """Two optimizer steps with the update chasing the backward per finished bucket (Trainer.chase_optimizer / C2W_CHASE_OPT=1: the
    all-reduce AND the fused AdamW + EMA of a bucket are issued while the rest of the backward still runs) and two with the update
    behind the whole backward, from the same initial weights, several buckets: the weights must be the same."""
    dev = _init(rank, world, port, backend)
    from climate2weather_amd.score import ScoreUNet
    from climate2weather_amd.training import Trainer

    g = np.load(os.path.join(golden_dir, "tiny_net.npz"))
    if world == 1:
        os.environ["C2W_FORCE_DIST"] = "1"
    x, t, eps = (torch.from_numpy(g[k]).to(dev) for k in ("x", "t", "eps"))
    per = x.shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)
    res = {}
    for chase in (False, True):
        torch.manual_seed(3)
        net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, embedding_dim=64, hidden_channels=[32, 64], hidden_blocks=[1, 1],
                        attention_levels=[1], kernel_size=3, padding_mode="zeros").to(dev)
        tr = Trainer(net, lr=1e-3, precision="fp32", ema_rates=[0.9], bucket_mb=0.05)
        tr.chase_optimizer = chase
        assert tr.sync_grads and len(tr.buckets) > 4
        calls = []
        orig = tr._update_range
        tr._update_range = lambda s, e, lr, step, _o=orig: (calls.append((s, e)), _o(s, e, lr, step))[1]
        for _ in range(2):
            loss = tr.step(x[sl].contiguous(), t=t[sl].reshape(-1), eps=eps[sl].contiguous())
        res[chase] = dict(loss=float(loss), sd={k: v.detach().cpu().clone() for k, v in net.state_dict().items()},
                          ema=tr.ema_flats[0].detach().cpu().clone(), update_calls=len(calls))
    torch.save(res, os.path.join(out_dir, f"chase{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def run_wire(rank: int, world: int, port: int, golden_dir: str, out_dir: str, backend: str = "gloo"):
    """The gradient all-reduce with bfloat16 on the wire (Trainer(allreduce_dtype="bf16"): every finished bucket cast, summed over the
    ranks in bf16, cast back) next to the fp32 wire, same weights and batches, several buckets: the summed gradients agree to bf16's
    resolution, the ranks stay in lock step, the loss is untouched (it is computed before any gradient moves)."""
    dev = _init(rank, world, port, backend)
    from climate2weather_amd.score import ScoreUNet
    from climate2weather_amd.training import Trainer

    g = np.load(os.path.join(golden_dir, "tiny_net.npz"))
    if world == 1:
        os.environ["C2W_FORCE_DIST"] = "1"
    x, t, eps = (torch.from_numpy(g[k]).to(dev) for k in ("x", "t", "eps"))
    per = x.shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)
    res = {}
    for wire in ("fp32", "bf16"):
        torch.manual_seed(3)
        net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, embedding_dim=64, hidden_channels=[32, 64], hidden_blocks=[1, 1],
                        attention_levels=[1], kernel_size=3, padding_mode="zeros").to(dev)
        tr = Trainer(net, lr=1e-3, precision="fp32", ema_rates=[0.9], bucket_mb=0.05, allreduce_dtype=wire)
        assert tr.sync_grads and len(tr.buckets) > 4 and (tr.wire is not None) == (wire == "bf16")
        n_ar, dts = [0], set()
        orig = dist.all_reduce

        def counting(tensor, *a, **k):
            n_ar[0] += 1
            dts.add(tensor.dtype)
            return orig(tensor, *a, **k)
        dist.all_reduce = counting
        try:
            loss = tr.step(x[sl].contiguous(), t=t[sl].reshape(-1), eps=eps[sl].contiguous())
        finally:
            dist.all_reduce = orig
        res[wire] = dict(loss=float(loss), grad=tr.eng.flat_grad.detach().cpu().clone(), flat=tr.eng.flat.detach().cpu().clone(),
                         all_reduces=n_ar[0], dtypes=sorted(str(d) for d in dts), nb=len(tr.buckets))
    torch.save(res, os.path.join(out_dir, f"wire{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def run_full_size(rank: int, world: int, port: int, out_dir: str, steps: int = 3, batch: int = 16):
    """The bench's own network (default configs/sda_unet.yml, C = 65, 128x128) in bf16 through a real RCCL communicator: every one of
    the 25-MB buckets of the 288-MB gradient buffer is all-reduced from the gradient stream while wgrad_patch_kernel / conv_patch
    launches of the same backward are still running.  Against the same steps without a process group (same seeds, same batches)."""
    dev = _init(rank, world, port, "nccl")
    from climate2weather_amd.score import ScoreUNet
    from climate2weather_amd.training import Trainer
    cfg = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3, padding_mode="zeros",
               attention_levels=[4])
    gen = torch.Generator().manual_seed(5 + rank)
    xs = [(torch.randn(batch, 65, 128, 128, generator=gen) * 0.5 + 0.5).to(dev) for _ in range(steps)]
    ts = [torch.rand(batch, generator=gen).to(dev) for _ in range(steps)]
    res = {}
    for mode in ("dist", "plain"):
        if mode == "dist":
            os.environ["C2W_FORCE_DIST"] = "1"
        else:
            os.environ.pop("C2W_FORCE_DIST", None)
        torch.manual_seed(0)
        net = ScoreUNet(channels=65, spatial=2, activation=torch.nn.SiLU, **cfg).to(dev)
        tr = Trainer(net, lr=1e-4, precision="bf16", ema_rates=[0.9999], seed=11)
        assert tr.sync_grads == (mode == "dist" or world > 1)
        n_ar = [0]
        orig = dist.all_reduce

        def counting(*a, **k):
            n_ar[0] += 1
            return orig(*a, **k)
        dist.all_reduce = counting
        try:
            losses = [float(tr.step(x, t=t)) for x, t in zip(xs, ts)]
        finally:
            dist.all_reduce = orig
        torch.cuda.synchronize()
        res[mode] = dict(losses=losses, flat=tr.eng.flat.detach().cpu().clone(), nb=len(tr.buckets), all_reduces=n_ar[0],
                         on_side_stream=tr._comm_stream() is not None)
        del tr, net
        torch.cuda.empty_cache()
    torch.save(res, os.path.join(out_dir, f"full{rank}.pt"))
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
    # more than one rank: the module's backward is the segmented chain (score.py::_GradSegment), so that DDP's reducer receives the
    # output side's gradients while the pass is still running -- the hooks below see them arrive in at least three instalments
    assert net._segments(list(net.parameters()), xt) == 8
    arrivals, launches = [], []
    real_wgrad = c2w_ops.conv_wgrad
    c2w_ops.conv_wgrad = lambda *a, **kw: (launches.append(1), real_wgrad(*a, **kw))[1]
    hooks = [p.register_hook(lambda g_, n=n: arrivals.append(len(launches))) for n, p in net.named_parameters()]
    loss = ((ddp(xt, t[sl].reshape(-1)) - eps[sl]) ** 2).mean()
    loss.backward()
    c2w_ops.conv_wgrad = real_wgrad
    for h in hooks:
        h.remove()
    assert len(arrivals) == len(list(net.parameters())) and len(set(arrivals)) >= 3 and arrivals[0] < len(launches), (sorted(set(arrivals)), len(launches))
    grads = {n: p.grad.clone() for n, p in net.named_parameters()}
    opt.step()
    torch.save(dict(loss=float(loss), grads=grads, sd={k: v.clone() for k, v in net.state_dict().items()}),
               os.path.join(out_dir, f"mod{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def run_five_strings_ddp(rank: int, world: int, port: int, golden_dir: str, out_dir: str, bucket_view: bool = False):
    """The reference's configuration: the module wrapped in torch DDP (fabric.setup_module, training_loop.py:116), the loop of
    training_loop.py:369-391, and all five seams pointing at this package -- in particular climate2weather_amd.optim.AdamW over the
    DDP-averaged gradients and SDAPipeline.loss as one autograd node called THROUGH the DDP wrapper.  Three steps; rank r sees item r
    of each global batch of 2.  Saved: losses, weights, whether the optimizer ran its flat path, and (rank 0) the same three steps in
    ONE process on the whole batch with torch.optim.AdamW and the reference's loss arithmetic for comparison."""
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
    c2w_ops.EMULATED = True
    from climate2weather_amd.ema import StandardEMA
    from climate2weather_amd.optim import AdamW
    from climate2weather_amd.pipelines import SDAPipeline
    from climate2weather_amd.score import ScoreUNet, _LossTensor

    cfg = dict(channels=6, spatial=2, activation=torch.nn.SiLU, embedding_dim=64, hidden_channels=[32, 64], hidden_blocks=[1, 1],
               attention_levels=[1], kernel_size=3, padding_mode="zeros")
    gen = torch.Generator().manual_seed(11)
    data = [torch.randn(world, 6, 16, 16, generator=gen) * 0.5 + 0.5 for _ in range(3)]
    draws = [(torch.rand(world, 1, 1, 1, generator=gen), torch.randn(world, 6, 16, 16, generator=gen)) for _ in range(3)]

    class Injected(SDAPipeline):  # the loop's own draws replaced by recorded ones, so that the ranks' items form one known batch
        fused_loss = "eps"

        def loss(self, net, x, forcing=None):
            from climate2weather_amd.pipelines import _engine_module
            core = _engine_module(net)
            t, eps = self._next
            core.__dict__["_loss_request"] = dict(eps=eps, eta=self.eta)
            try:
                return net(x, t)
            finally:
                core.__dict__.pop("_loss_request", None)

    torch.manual_seed(3)
    net = ScoreUNet(**cfg)
    # bucket_view: p.grad become views of the reducer's buckets after the first step -- the drop-in AdamW gathers them and keeps its fused step
    ddp = torch.nn.parallel.DistributedDataParallel(net, gradient_as_bucket_view=bucket_view)
    pipeline, optimizer, ema = Injected(), AdamW(params=net.parameters(), lr=1e-3, weight_decay=1e-3, betas=[0.9, 0.999]), StandardEMA(net=net)
    losses, fused = [], []
    for i in range(3):
        optimizer.zero_grad()
        pipeline._next = (draws[i][0][rank:rank + 1], draws[i][1][rank:rank + 1])
        out = pipeline.loss(net=ddp, x=data[i][rank:rank + 1])
        fused.append(isinstance(out, _LossTensor))
        loss = out.mean().mul(1.0)
        loss.backward()
        optimizer.step()
        losses.append(loss.detach().item())
        ema.update()
    res = dict(losses=losses, sd={k: v.detach().clone() for k, v in net.state_dict().items()}, flat=optimizer.fused_path_active(), fused=fused,
               ema={k: v.detach().clone() for k, v in ema.emas[0].state_dict().items()})
    if rank == 0:  # single process, whole batch, torch's optimizer, the reference's tensor arithmetic
        torch.manual_seed(3)
        ref = ScoreUNet(**cfg)
        opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=1e-3, betas=(0.9, 0.999))
        plain = SDAPipeline()
        ref_losses = []
        for i in range(3):
            opt.zero_grad()
            t, eps = draws[i]
            xt = plain.mu(t) * data[i] + plain.sigma(t) * eps
            l = ((ref(xt, t) - eps) ** 2).mean()
            l.backward()
            opt.step()
            ref_losses.append(l.item())
        res["ref_losses"], res["ref_sd"] = ref_losses, {k: v.detach().clone() for k, v in ref.state_dict().items()}
    torch.save(res, os.path.join(out_dir, f"five{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def run_module_ddp_gpu(rank: int, world: int, port: int, out_dir: str):
    """The drop-in path under torch's DistributedDataParallel over RCCL on the HIP engine: the five-string loop in bf16 autocast, the
    module's backward as the segmented chain (score.py::_GradSegment; forced on for one rank), three steps -- next to the same three
    steps without a process group and with the single node.  What this can show on one GPU: DDP's reducer (bucket copies on the
    caller's stream, all-reduce on its own) reads gradients that the engine's gradient stream wrote, ordered by the chain's events."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from climate2weather_amd import ops as c2w_ops
    from climate2weather_amd.ema import StandardEMA
    from climate2weather_amd.optim import AdamW
    from climate2weather_amd.pipelines import SDAPipeline
    from climate2weather_amd.score import ScoreUNet

    cfg = dict(channels=6, spatial=2, activation=torch.nn.SiLU, embedding_dim=64, hidden_channels=[64, 128, 128], hidden_blocks=[1, 1, 1],
               attention_levels=[2], kernel_size=3, padding_mode="zeros")
    gen = torch.Generator().manual_seed(11)
    data = [(torch.randn(4 * world, 6, 32, 32, generator=gen) * 0.5 + 0.5) for _ in range(3)]

    def loop(use_ddp: bool):
        torch.manual_seed(3)
        net = ScoreUNet(**cfg).to(dev)
        net.grad_segments = 4 if use_ddp else 1
        mod = torch.nn.parallel.DistributedDataParallel(net, device_ids=[rank]) if use_ddp else net
        pipeline, optimizer, ema = SDAPipeline(), AdamW(params=net.parameters(), lr=1e-3, weight_decay=1e-3, betas=[0.9, 0.999]), StandardEMA(net=net)
        arrivals, launches = [], []
        real = c2w_ops.conv_wgrad
        c2w_ops.conv_wgrad = lambda *a, **kw: (launches.append(1), real(*a, **kw))[1]
        hooks = [p.register_hook(lambda g_: arrivals.append(len(launches))) for p in net.parameters()]
        losses = []
        try:
            for i in range(3):
                optimizer.zero_grad()
                torch.manual_seed(100 + i)  # the loss draws t and the noise seed from torch's CPU stream
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = pipeline.loss(net=mod, x=data[i][4 * rank: 4 * rank + 4].to(dev)).mean().mul(1.0)
                loss.backward()
                optimizer.step()
                losses.append(loss.detach().item())
                ema.update()
        finally:
            c2w_ops.conv_wgrad = real
            for h in hooks:
                h.remove()
        torch.cuda.synchronize()
        n_par = len(list(net.parameters()))
        return dict(losses=losses, flat=net._get_engine().flat.detach().clone().cpu(), flat_path=optimizer.fused_path_active(),
                    arrivals=arrivals[:n_par], launches=len(launches) // 3, ema=ema.emas[0]._get_engine().flat.detach().clone().cpu())
    res = dict(ddp=loop(True), world=world)
    if world == 1:
        res["plain"] = loop(False)
    torch.save(res, os.path.join(out_dir, f"modgpu{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
