"""Ensemble sampling driver: the inner loops of ``exp/downscaling.py:208-265`` without the xarray/netCDF I/O around
them.  Members are sharded across ranks exactly as the reference does (``num_samples % world == 0``; rank r generates
members ``r*n .. (r+1)*n - 1``; no collective); every member's trajectory lives in HBM for the whole run.
``members_per_batch`` > 1 co-samples that many of a rank's members: their windows share the network batches, which is
what fills an MI355X at the shipped trajectory lengths (L = 49: 37 windows per member; 288 GB of HBM hold hundreds of
members' states and activations).  With ``corrections == 0`` (the shipped configuration) the members are the ones the
one-by-one loop produces from the same seed -- and co-sampling up to the score function's window floor is the DEFAULT
(``members_per_batch=None``); with corrections the corrector normals are drawn in another order, so the default stays 1.

Random numbers (``rng``): "reference" (default) draws them where the reference draws them -- ``set_random_seed(seed, rank)``
seeds torch's CPU generator (exp/downscaling.py:100-103, util.py:27-29), every member's initial noise is one
``torch.randn(L, C, H, W)`` on that generator in member order (exp/downscaling.py:250) and the corrector normals are CPU draws
in the order ``SDAPipeline.sample`` makes them (its state lives on the CPU there: ``z.normal_()``, src/thor/pipelines.py:59-60,82) -- so seed s,
rank r, member i is the SAME member as the reference's, to the network's arithmetic.  "device" draws on the GPU generator
instead (no host-side normals, no H2D copy: 0.57 G normals per corrector step at L = 8737), a different but equally valid stream."""
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
                 on_sample: Optional[Callable[[int, torch.Tensor], None]] = None, show_progressbar: bool = False,
                 members_per_batch: Optional[int] = None, rng: str = "reference") -> List[Tuple[int, torch.Tensor]]:
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    assert num_samples % world == 0, "Number of samples must be divisible by the number of devices."  # exp/downscaling.py:96-98
    per_gpu = num_samples // world
    pipeline = pipeline or SDAPipeline()
    device = torch.device(device) if device is not None else next(net.parameters()).device
    if precision is not None:
        net.precision = precision
    net.eval()
    if rng not in ("reference", "device"):
        raise ValueError(f"rng must be 'reference' or 'device', got {rng!r}")
    set_random_seed(seed, rank)  # exp/downscaling.py:100-103: members differ across ranks through the seed
    shape = (length, n_vars, height, width)

    def draw():
        """One (L, C, H, W) standard-normal field from the stream the mode names."""
        return torch.randn(shape).to(device) if rng == "reference" else torch.randn(shape, device=device)

    def corrector_draws(n_members: int):
        """The corrector normals in the order the sampler consumes them (one per correction per step), or None: the sampler
        then draws on the state's device."""
        if corrections == 0 or rng == "device":
            return None
        return ((draw() if n_members == 1 else torch.stack([draw() for _ in range(n_members)], 0)) for _ in range(steps * corrections))
    score_fn = BatchedScoreFunction(net, markov_order=markov_order, batch_size=batch_size, device=device, noise_process=pipeline)
    if A is not None:
        score_fn.condition_on(A=A, y=y, std=std, gamma=gamma, exact_grad=exact_grad)
    out = []
    if members_per_batch is None:
        # Default: without a corrector (every shipped configuration, exp/configs/**) co-sampled members are the one-by-one members -- the
        # same noise, the same arithmetic per window; in fp32 equal to 1e-4 (tested), in the 16-bit modes equal to ROUNDING only, because
        # the kernel a layer runs on depends on the number of windows in the batch (tests/test_gpu_host.py: bf16 tolerance test) -- so as
        # many of this rank's members share the network batches as it takes to reach the score function's window floor (L = 49: 37
        # windows per member, 7 members; 6.4 k -> 9.2 k window-forwards/s).  With a corrector the normals would be drawn in another
        # order than the reference's loop draws them: one member at a time unless the caller asks.
        # Observable differences from the reference's loop (exp/downscaling.py:248-265): on_sample fires for a group's members when the
        # whole group has finished, and a NaN in ONE member (raised one step late, pipelines.HostRing) discards the group it was
        # sampled with; members_per_batch=1 restores the reference's behaviour exactly.
        nwin = max(1, length - 2 * markov_order)
        floor = score_fn._window_floor(height * width) if corrections == 0 else 1
        group = min(per_gpu, max(1, -(-floor // nwin)))
    else:
        group = max(1, int(members_per_batch))
    for i0 in range(0, per_gpu, group):
        ids = [rank * per_gpu + i for i in range(i0, min(i0 + group, per_gpu))]
        noises = [draw() for _ in ids]  # one draw per member, in member order (exp/downscaling.py:248-250)
        if len(ids) == 1:
            xs = [pipeline.sample(score_fn, noises[0], steps=steps, corrections=corrections, tau=tau, device=device,
                                  show_progressbar=show_progressbar, z_draws=corrector_draws(1))]
        else:
            xs = list(pipeline.sample(score_fn, torch.stack(noises, 0), steps=steps, corrections=corrections, tau=tau, device=device,
                                      show_progressbar=show_progressbar, z_draws=corrector_draws(len(ids))))
        for sample_id, x in zip(ids, xs):
            if on_sample is not None:
                on_sample(sample_id, x)
            else:
                out.append((sample_id, x))
    return out
