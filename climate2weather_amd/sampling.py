"""Ensemble sampling driver: the inner loops of ``exp/downscaling.py:208-265`` without the xarray/netCDF I/O around
them.  Members are sharded across ranks exactly as the reference does (``num_samples % world == 0``; rank r generates
members ``r*n .. (r+1)*n - 1``; no collective); every member's trajectory lives in HBM for the whole run."""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

import torch

from .pipelines import SDAPipeline
from .score_fn import BatchedScoreFunction
from .util import set_random_seed


def run_ensemble(net, pipeline: Optional[SDAPipeline] = None, *, length: int, n_vars: int, height: int, width: int, markov_order: int,
                 num_samples: int, steps: int = 256, corrections: int = 0, tau: float = 0.5, batch_size: int = 128,
                 A=None, y=None, std=None, gamma: float = 1e-2, exact_grad: bool = False, seed: int = 0, rank: Optional[int] = None,
                 world: Optional[int] = None, device=None, precision: Optional[str] = "bf16",
                 on_sample: Optional[Callable[[int, torch.Tensor], None]] = None, show_progressbar: bool = False) -> List[Tuple[int, torch.Tensor]]:
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    assert num_samples % world == 0, "Number of samples must be divisible by the number of devices."  # exp/downscaling.py:96-98
    per_gpu = num_samples // world
    pipeline = pipeline or SDAPipeline()
    device = torch.device(device) if device is not None else next(net.parameters()).device
    if precision is not None:
        net.precision = precision
    net.eval()
    set_random_seed(seed, rank)  # exp/downscaling.py:100-103: members differ across ranks through the seed
    score_fn = BatchedScoreFunction(net, markov_order=markov_order, batch_size=batch_size, device=device, noise_process=pipeline)
    if A is not None:
        score_fn.condition_on(A=A, y=y, std=std, gamma=gamma, exact_grad=exact_grad)
    out = []
    for i in range(per_gpu):
        sample_id = rank * per_gpu + i
        noise = torch.randn(length, n_vars, height, width, device=device)
        x = pipeline.sample(score_fn, noise, steps=steps, corrections=corrections, tau=tau, device=device,
                            show_progressbar=show_progressbar)
        if on_sample is not None:
            on_sample(sample_id, x)
        else:
            out.append((sample_id, x))
    return out
