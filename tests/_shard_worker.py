"""Worker for the world_size-2 tests of the time-sharded sampler: "gloo" on the CPU (HIP launchers replaced by tests/emu_ops) and
"nccl" = RCCL on two GPUs (the real kernels, halo frames point-to-point over xGMI)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist


def run(rank: int, world: int, port: int, golden_dir: str, out_dir: str, backend: str = "gloo"):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _ddp_worker import _init
    dev = _init(rank, world, port, backend)
    from climate2weather_amd.pipelines import SDAPipeline
    from climate2weather_amd.score import ScoreUNet
    from climate2weather_amd.score_fn import PoolStrideOperator
    from climate2weather_amd.sharded import TimeShardedScoreFunction, sample_time_sharded

    s = {k: v for k, v in np.load(os.path.join(golden_dir, "sampler.npz")).items()}
    sg = np.load(os.path.join(golden_dir, "sampler_gamma.npz"))
    s["cond_c0_gvec.noise"], s["cond_c0_gvec.x"] = s["cond_c0.noise"], sg["cond_c0_gvec.x"]
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, embedding_dim=64, hidden_channels=[32, 64], hidden_blocks=[1, 1],
                    attention_levels=[1], kernel_size=3, padding_mode="zeros").to(dev).eval()
    net.precision = "fp32"
    pipe = SDAPipeline()
    out = {}
    for name, corrections, cond in [("uncond_c0", 0, False), ("uncond_c1", 1, False), ("cond_c0", 0, True), ("cond_c0_gvec", 0, True),
                                    ("cond_c1_exact", 1, True)]:  # exact gradient: reverse halo exchange of dlog p/dx
        noise = torch.from_numpy(s[name + ".noise"])
        sf = TimeShardedScoreFunction(net, markov_order=1, length=noise.shape[0], batch_size=3, device=dev,
                                      noise_process=pipe)
        if cond:
            sf.condition_on(A=PoolStrideOperator(8, 2), y=torch.from_numpy(s["y_obs"]), std=torch.from_numpy(s["std"]),
                            gamma=torch.from_numpy(sg["gamma"]) if name.endswith("_gvec") else float(s["gamma"]),  # (1, F, 1, 1): exp/downscaling.py:228-233
                            exact_grad=name.endswith("_exact"))
        lo, hi = sf.bounds[rank]
        zs = [torch.from_numpy(z)[lo:hi] for z in s[name + ".z"]] if corrections else None
        x = sample_time_sharded(pipe, sf, noise[lo:hi], steps=4, corrections=corrections, tau=0.5, z_draws=zs, gather=True)
        out[name] = x.cpu().clone()
        out[name + ".bounds"] = sf.bounds
        if name == "cond_c0":  # the same run with the halo exchange waited for BEFORE any window runs (no overlap): same trajectory
            sf.overlap_halo = False
            x2 = sample_time_sharded(pipe, sf, noise[lo:hi], steps=4, corrections=corrections, tau=0.5, z_draws=zs, gather=True)
            out[name + ".no_overlap"] = x2.cpu().clone()
    torch.save(out, os.path.join(out_dir, f"shard{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
