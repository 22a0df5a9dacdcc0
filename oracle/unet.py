"""CPU oracle for the score network -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional restatement (plain PyTorch fp32 ops over a flat ``state_dict``) of
the reference's ``ScoreUNet`` so that the HIP path can be checked against it on
machines where ``/root/reference`` does not exist.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.

Pinned against the imported reference by ``tests/golden/make_golden.py`` (run in
the build container) -> ``tests/golden/*.npz`` -> ``tests/test_oracle.py``.

PARITY UNPINNED at one boundary: the channel LayerNorm is the un-vendored
``zuko==1.0.1`` (reference requirements.txt:33).  The reference carries no tests
and zuko is not installed, so "unbiased variance, eps inside the sqrt, no
affine" is recalled, not verified.  ``ln_unbiased`` is the
single switch for it.

Each function cites the reference file:line it follows.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor

SD = Dict[str, Tensor]


def timestep_embedding(t: Tensor, dim: int = 32, max_period: float = 10000.0) -> Tensor:
    """model/score.py:14-34 -- [cos(t f) | sin(t f)], f_i = exp(-ln(max_period) i / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half).to(t.device)
    ang = t.reshape(-1, 1).float() * freqs.reshape(1, -1)
    emb = torch.cat((ang.cos(), ang.sin()), dim=-1)
    if dim % 2:
        emb = torch.cat((emb, torch.zeros_like(emb[:, :1])), dim=-1)
    return emb.to(t.dtype)


def channel_layer_norm(x: Tensor, dim: int = 1, eps: float = 1e-5, ln_unbiased: bool = True) -> Tensor:
    """zuko.nn.LayerNorm as used at model/nn.py:44,154,183 (third-party; see header)."""
    var, mean = torch.var_mean(x, dim=dim, keepdim=True, unbiased=ln_unbiased)
    return (x - mean) / (var + eps).sqrt()


def time_mlp(sd: SD, t: Tensor, noise_features: int = 32, forcing: Optional[Tensor] = None) -> Tensor:
    """model/score.py:61-67; with ``forcing`` the projection of model/score.py:49-51 is added before the second SiLU (:65-66)."""
    e = timestep_embedding(t.reshape(-1), noise_features)
    e = F.silu(F.linear(e, sd["map_layer0.weight"], sd["map_layer0.bias"]))
    e = F.linear(e, sd["map_layer1.weight"], sd["map_layer1.bias"])
    if forcing is not None:
        e = e + F.linear(forcing, sd["map_forcing.weight"], sd["map_forcing.bias"])
    return F.silu(e)


def mod_res_block(sd: SD, p: str, x: Tensor, emb: Tensor, act, ln_unbiased: bool = True) -> Tensor:
    """model/nn.py:27-28 with the residue built at model/nn.py:146-159."""
    m = F.linear(emb, sd[p + "project.0.weight"], sd[p + "project.0.bias"])
    h = channel_layer_norm(x + m[:, :, None, None], 1, ln_unbiased=ln_unbiased)
    h = F.conv2d(h, sd[p + "residue.1.weight"], sd[p + "residue.1.bias"], padding=1)
    h = act(h)
    h = F.conv2d(h, sd[p + "residue.3.weight"], sd[p + "residue.3.bias"], padding=1)
    return x + h


def attention_block(sd: SD, p: str, x: Tensor, ln_unbiased: bool = True) -> Tensor:
    """model/nn.py:49-59 and QKVAttention model/nn.py:67-85 (one head)."""
    b, c = x.shape[:2]
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(channel_layer_norm(xf, 1, ln_unbiased=ln_unbiased), sd[p + "qkv.weight"], sd[p + "qkv.bias"])
    q, k, v = qkv.split(c, dim=1)
    s = 1.0 / math.sqrt(math.sqrt(c))
    w = torch.einsum("bct,bcs->bts", q * s, k * s)
    w = torch.softmax(w.float(), dim=-1).to(w.dtype)
    h = torch.einsum("bts,bcs->bct", w, v)
    h = F.conv1d(h, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return (xf + h).reshape(x.shape)


def unet_forward(
    sd: SD,
    x: Tensor,
    emb: Tensor,
    hidden_blocks: Sequence[int],
    attention_levels: Sequence[int] = (),
    act=F.silu,
    ln_unbiased: bool = True,
    prefix: str = "unet.",
) -> Tensor:
    """model/nn.py:220-242.  ``tails``/``ascent`` are stored reversed (model/nn.py:216,218):
    key index j holds level L-1-j."""
    L = len(hidden_blocks)

    def run_blocks(stem: str, level: int, x: Tensor) -> Tensor:
        per = 2 if level in attention_levels else 1
        for bi in range(hidden_blocks[level]):
            x = mod_res_block(sd, f"{stem}.{bi * per}.", x, emb, act, ln_unbiased)
            if per == 2:
                x = attention_block(sd, f"{stem}.{bi * per + 1}.", x, ln_unbiased)
        return x

    skips = []
    for i in range(L):
        if i == 0:
            x = F.conv2d(x, sd[prefix + "heads.0.weight"], sd[prefix + "heads.0.bias"], padding=1)
        else:
            x = F.conv2d(x, sd[f"{prefix}heads.{i}.0.weight"], sd[f"{prefix}heads.{i}.0.bias"], stride=2, padding=1)
        x = run_blocks(f"{prefix}descent.{i}", i, x)
        skips.append(x)
    skips.pop()
    for j in range(L):
        level = L - 1 - j
        x = run_blocks(f"{prefix}ascent.{j}", level, x)
        if level > 0:
            h = channel_layer_norm(x, 1, ln_unbiased=ln_unbiased)
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"{prefix}tails.{j}.2.weight"], sd[f"{prefix}tails.{j}.2.bias"], padding=1)
            x = h + skips.pop()
        else:
            x = F.conv2d(x, sd[f"{prefix}tails.{j}.weight"], sd[f"{prefix}tails.{j}.bias"], padding=1)
    return x


def score_unet_forward(
    sd: SD,
    x: Tensor,
    t: Tensor,
    hidden_blocks: Sequence[int],
    attention_levels: Sequence[int] = (),
    act=F.silu,
    ln_unbiased: bool = True,
    forcing: Optional[Tensor] = None,
) -> Tensor:
    """model/score.py:59-70."""
    emb = time_mlp(sd, t, forcing=forcing)
    return unet_forward(sd, x, emb, hidden_blocks, attention_levels, act, ln_unbiased).reshape(x.shape)


class OracleScoreUNet(torch.nn.Module):
    """nn.Module face of the oracle (same ctor keywords as model/score.py:46 /
    model/nn.py:108) so the oracle diffusion glue can call ``net(x, t)``.  Holds a
    plain ParameterDict keyed by the reference's state_dict names."""

    def __init__(self, state_dict: SD, hidden_blocks, attention_levels=(), act=F.silu, ln_unbiased=True):
        super().__init__()
        self.keys = list(state_dict.keys())
        self.params = torch.nn.ParameterList([torch.nn.Parameter(v.detach().clone().float()) for v in state_dict.values()])
        self.hidden_blocks = list(hidden_blocks)
        self.attention_levels = list(attention_levels)
        self.act = act
        self.ln_unbiased = ln_unbiased

    def sd(self) -> SD:
        return dict(zip(self.keys, self.params))

    def forward(self, x: Tensor, t: Tensor, forcing=None) -> Tensor:
        return score_unet_forward(self.sd(), x, t, self.hidden_blocks, self.attention_levels, self.act, self.ln_unbiased, forcing=forcing)
