"""Quantile (de)normalisation either side of the sampler (``data/pipeline.py:183-244`` applied at
``exp/downscaling.py:150,188,198,274``).  The reference does it in xarray on the host; every mode is a per-variable
affine map, so on an ``(L, F, H, W)`` trajectory that already lives in HBM it is one streaming kernel
(``c2w_affine_channels``).  netCDF / quantile-file I/O stays with the caller: pass the quantile values."""
from __future__ import annotations

from typing import Dict, Sequence

import torch

from . import ops

MODES = {  # mode -> (quantile level subtracted, range from level, range to level)
    "minmax": (0.0, 0.0, 1.0),
    "robust": (0.5, 0.25, 0.75),
    "robust95": (0.5, 0.05, 0.95),
    "quant95": (0.05, 0.05, 0.95),
    "quant99": (0.01, 0.01, 0.99),
}


class QuantileNormalizer:
    """``quantiles``: {level: per-variable values} in the order of the trajectory's variable axis (the reference sorts
    ``data_vars``, ``exp/downscaling.py:100``).  ``normalize`` / ``unnormalize`` mirror ``normalize_ds`` / ``unnormalize_ds``."""

    def __init__(self, quantiles: Dict[float, Sequence[float]], mode: str = "quant95"):
        if mode not in MODES:
            raise ValueError(f"Invalid mode: {mode}")  # data/pipeline.py:212
        sub, lo, hi = MODES[mode]
        self.mode = mode
        self.lower = torch.as_tensor(quantiles[sub], dtype=torch.float64).reshape(-1)
        self.range = (torch.as_tensor(quantiles[hi], dtype=torch.float64) - torch.as_tensor(quantiles[lo], dtype=torch.float64)).reshape(-1)
        self._dev = {}

    def _coef(self, device, inverse: bool):
        key = (str(device), inverse)
        if key not in self._dev:
            if inverse:  # x * range + lower
                scale, shift = self.range, self.lower
            else:        # (x - lower) / range
                scale, shift = 1.0 / self.range, -self.lower / self.range
            self._dev[key] = (scale.float().to(device).contiguous(), shift.float().to(device).contiguous())
        return self._dev[key]

    def _apply(self, x: torch.Tensor, inverse: bool, out=None) -> torch.Tensor:
        if x.dim() < 3 or x.shape[-3] != self.lower.numel():
            raise ValueError(f"expected (..., {self.lower.numel()}, H, W), got {tuple(x.shape)}")
        x = x.float().contiguous()
        F, HW = x.shape[-3], x.shape[-1] * x.shape[-2]
        y = torch.empty_like(x) if out is None else out
        scale, shift = self._coef(x.device, inverse)
        ops.affine_channels(x, y, scale, shift, x.numel() // HW, F, HW)
        return y

    def normalize(self, x: torch.Tensor, out=None) -> torch.Tensor:
        return self._apply(x, False, out)

    def unnormalize(self, x: torch.Tensor, out=None) -> torch.Tensor:
        return self._apply(x, True, out)
