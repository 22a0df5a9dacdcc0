"""Stand-in for the one symbol the reference imports from the un-vendored
third-party package `zuko==1.0.1` (reference requirements.txt:33; used at
model/nn.py:8,44,154,183).

TEST INFRASTRUCTURE ONLY. It exists so that `tests/golden/make_golden.py` can
import the reference's own `model/nn.py` in the build container.  It never
travels into the product path.

PARITY UNPINNED at this boundary: zuko is not installed here and the reference
holds no tests, so the definition below is *recalled* from zuko 1.0.x
(`torch.var_mean` with its default unbiased estimator, eps inside the sqrt, no
affine parameters), not verified against the package.
"""
from typing import Sequence, Union

import torch
from torch import Tensor


class LayerNorm(torch.nn.Module):
    def __init__(self, dim: Union[int, Sequence[int]] = -1, eps: float = 1e-5):
        super().__init__()
        self.dim = dim if isinstance(dim, int) else tuple(dim)
        # zuko registers eps as a buffer; keep it non-persistent so the
        # state_dict stays at the 228 parameter tensors SURVEY.md A1 lists.
        self.register_buffer("eps", torch.as_tensor(eps), persistent=False)

    def forward(self, x: Tensor) -> Tensor:
        variance, mean = torch.var_mean(x, dim=self.dim, keepdim=True)
        return (x - mean) / (variance + self.eps).sqrt()
