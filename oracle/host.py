"""CPU oracle for the host-side pieces around the training step -- TEST
INFRASTRUCTURE, NOT PRODUCT CODE.  (training_loop.py, dataset.py and util.py are
not importable here -- lightning/h5py/torchvision are absent -- so these follow
the source text; the known-answer values in tests/golden/kat_host.json were
computed by running the quoted expressions, see make_golden.py.)
"""
from __future__ import annotations

from typing import Iterator, List

import numpy as np
import torch


def seed_from_args(*args) -> int:
    """util.py:27-29: ``hash(args) % (1 << 31)`` (CPython int-tuple hash is unsalted)."""
    return hash(args) % (1 << 31)


def seed_everything(seed: int) -> None:
    """What ``lightning.fabric.seed_everything(seed, workers=True)`` seeds (util.py:27-29 calls it; lightning 2.2.1 is not
    installed here, its documented behaviour: python ``random``, numpy's global state and ``torch.manual_seed`` -- which also
    seeds every CUDA generator)."""
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def ensemble_members(sample_member, *, seed: int, rank: int, world: int, num_samples: int, shape, steps: int, corrections: int = 0):
    """The member loop of exp/downscaling.py:96-103,248-265 for ONE rank: ``num_samples % world == 0``; the process is seeded
    with ``hash((seed, rank)) % 2**31``; member i of the rank has ``sample_id = rank * n_per_gpu + i`` and starts from ONE
    ``torch.randn(L, C, H, W)`` drawn on the CPU generator; the sampler then draws its corrector normals from the same generator
    (``pipeline.sample`` keeps its state on the CPU in the reference and fills ``z.normal_()``, src/thor/pipelines.py:59-60,70-82: the
    same stream as ``torch.randn`` of that shape), one per
    correction per step, before the next member's noise.
    ``sample_member(noise, z_draws) -> x`` runs the sampler with those draws (oracle.diffusion.sample takes them explicitly).
    -> [(sample_id, x)]"""
    assert num_samples % world == 0, "num_samples must be divisible by world_size"
    per_gpu = num_samples // world
    seed_everything(seed_from_args(seed, rank))
    out = []
    for i in range(per_gpu):
        sample_id = rank * per_gpu + i
        noise = torch.randn(*shape)
        zs = [torch.randn(*shape) for _ in range(steps * corrections)]
        out.append((sample_id, sample_member(noise, zs)))
    return out


def linear_lr(cur_ndata: int, total_ndata: int, ref_lr: float) -> float:
    """src/thor/lr.py:17-19."""
    return ref_lr * (1 - cur_ndata / total_ndata)


def infinite_order(dataset_size: int, rank: int, num_replicas: int, seed: int, start_idx: int, count: int,
                   shuffle: bool = True) -> List[int]:
    """dataset.py:23-40: rank-strided walk over per-epoch permutations; first ``count`` indices."""
    out: List[int] = []
    idx = start_idx + rank
    epoch = None
    order = None
    while len(out) < count:
        if epoch != idx // dataset_size:
            epoch = idx // dataset_size
            order = np.arange(dataset_size)
            if shuffle:
                np.random.RandomState(hash((seed, epoch)) % (1 << 31)).shuffle(order)
        out.append(int(order[idx % dataset_size]))
        idx += num_replicas
    return out


def window_item(data: torch.Tensor, i: int, window: int) -> torch.Tensor:
    """dataset.py:114-126: item i = frames i..i+w-1 of x[N,C,H,W] flattened to (w*C,H,W), channel = tau*C+c."""
    return data[i : i + window].reshape(-1, *data.shape[2:])


def ema_update(p_ema: torch.Tensor, p_net: torch.Tensor, rate: float) -> torch.Tensor:
    """src/thor/ema.py:23-27: p_ema <- rate * p_ema + (1 - rate) * p_net."""
    return p_ema.mul(rate).add(p_net, alpha=1 - rate)


def adamw_step(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-3):
    """torch.optim.AdamW single-tensor rule as configured at train.py:176-181 (decoupled decay first)."""
    p = p * (1 - lr * weight_decay)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1**step
    bc2 = 1 - beta2**step
    denom = (v.sqrt() / (bc2**0.5)) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


def batch_split(batch_size: int, world_size: int, batch_gpu=None):
    """training_loop.py:57-62: per-rank batch and accumulation rounds."""
    total = batch_size // world_size
    if batch_gpu is None or batch_gpu > total:
        batch_gpu = total
    rounds = total // batch_gpu
    assert batch_size == batch_gpu * rounds * world_size
    return batch_gpu, rounds


# ---------------------------------------------------------------- quantile (de)normalisation, measurement operator
_NORM_MODES = {  # data/pipeline.py:183-244: every mode is (x - lower) / range with these quantile levels
    "minmax": (0.0, 0.0, 1.0),      # (level subtracted, range from, range to)
    "robust": (0.5, 0.25, 0.75),
    "robust95": (0.5, 0.05, 0.95),
    "quant95": (0.05, 0.05, 0.95),
    "quant99": (0.01, 0.01, 0.99),
}


def norm_coefficients(quantiles: dict, mode: str):
    """quantiles: {level: [value per variable]} -> (lower, range) per variable (data/pipeline.py:189-213)."""
    if mode not in _NORM_MODES:
        raise ValueError(f"Invalid mode: {mode}")
    sub, lo, hi = _NORM_MODES[mode]
    lower = torch.as_tensor(quantiles[sub], dtype=torch.float64)
    rng = torch.as_tensor(quantiles[hi], dtype=torch.float64) - torch.as_tensor(quantiles[lo], dtype=torch.float64)
    return lower, rng


def normalize(x, quantiles: dict, mode: str):
    """data/pipeline.py:183-213 on an (L, F, H, W) array: (x - lower_f) / range_f per variable f."""
    lower, rng = norm_coefficients(quantiles, mode)
    return ((x.double() - lower.view(1, -1, 1, 1)) / rng.view(1, -1, 1, 1)).to(x.dtype)


def unnormalize(x, quantiles: dict, mode: str):
    """data/pipeline.py:216-244: x * range_f + lower_f."""
    lower, rng = norm_coefficients(quantiles, mode)
    return (x.double() * rng.view(1, -1, 1, 1) + lower.view(1, -1, 1, 1)).to(x.dtype)


def measure(x, s_step: int, t_step: int):
    """exp/downscaling.py:129-132: AvgPool2d(s_step, stride=s_step)(x[::t_step]) on (L, F, H, W)."""
    return torch.nn.functional.avg_pool2d(x[::t_step], s_step, stride=s_step, padding=0)
