"""Parameter containers mirroring the reference's ``model/nn.py`` module tree.

The classes here own *parameters only* -- names, shapes and creation order match the reference
(model/nn.py:108-218) so a reference ``state_dict`` loads unchanged and
``torch.manual_seed(s)`` gives bit-identical initial weights.  All arithmetic is done by the HIP engine
(engine.py); none of these modules' ``forward`` is on the product path.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Union

import torch


class ModResidualBlock(torch.nn.Module):
    """Parameter holder for model/nn.py:18-28: ``project`` = Linear(mod -> C), ``residue`` = [LN, conv, act, conv]."""

    def __init__(self, project: torch.nn.Module, residue: torch.nn.Module):
        super().__init__()
        self.project = project
        self.residue = residue


class AttentionBlock(torch.nn.Module):
    """Parameter holder for model/nn.py:31-59 (single head)."""

    def __init__(self, channels: int, num_heads: int = 1):
        super().__init__()
        if num_heads != 1:
            raise NotImplementedError("climate2weather_amd: attention with num_heads != 1 is not built (the reference's UNet only ever constructs "
                                      "AttentionBlock(channels) with the default single head, model/nn.py:203-204; its multi-head reshape is "
                                      "model/nn.py:62-85).  INTEGRATION.md section 'What raises' lists every such fence.")
        self.channels = channels
        self.num_heads = num_heads
        self.qkv = torch.nn.Conv1d(channels, channels * 3, kernel_size=1)
        self.proj_out = torch.nn.Conv1d(channels, channels, kernel_size=1)


@dataclass
class BlockSpec:
    kind: str  # "res" | "attn"
    key: str  # state_dict prefix, e.g. "descent.4.1"
    channels: int
    mod_offset: int = -1  # res blocks: offset of this block's slice in the concatenated modulation vector


@dataclass
class LevelSpec:
    channels: int
    head_key: str
    tail_key: str
    descent: List[BlockSpec] = field(default_factory=list)
    ascent: List[BlockSpec] = field(default_factory=list)


class UNet(torch.nn.Module):
    """Same constructor as the reference ``model.nn.UNet`` (model/nn.py:108-121)."""

    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        mod_features: int,
        hidden_channels: Sequence[int] = (32, 64, 128),
        hidden_blocks: Sequence[int] = (2, 3, 5),
        attention_levels: Sequence[int] = (),
        kernel_size: Union[int, Sequence[int]] = 3,
        stride: Union[int, Sequence[int]] = 2,
        activation: Callable[[], torch.nn.Module] = torch.nn.ReLU,
        spatial: int = 2,
        **kwargs,
    ):
        super().__init__()
        ks = [kernel_size] * spatial if isinstance(kernel_size, int) else list(kernel_size)
        st = [stride] * spatial if isinstance(stride, int) else list(stride)
        if spatial != 2 or ks != [3, 3] or st != [2, 2]:
            raise NotImplementedError(f"climate2weather_amd: spatial={spatial}, kernel_size={ks}, stride={st} -- the kernels cover what every shipped "
                                      "config builds (configs/sda_unet.yml, train.py:164-173: spatial=2, kernel_size=3, stride=2); the reference's "
                                      "generic N-d / any-kernel construction (model/nn.py:126-143) has no MI355X path.  Use the reference module "
                                      "for such a network (INTEGRATION.md, 'What raises').")
        if kwargs.get("padding_mode", "zeros") != "zeros":
            raise NotImplementedError("climate2weather_amd: only padding_mode='zeros' (configs/sda_unet.yml:14; the halo-patch kernels realise the "
                                      "padding as out-of-range buffer loads); other modes of model/nn.py:126-143 need the reference module "
                                      "(INTEGRATION.md, 'What raises').")
        act = activation()
        if isinstance(act, torch.nn.SiLU):
            self.activation_kind = "silu"   # what train.py:171 passes
        elif isinstance(act, torch.nn.ReLU):
            self.activation_kind = "relu"   # the reference's own default (model/nn.py:118)
        else:
            raise NotImplementedError(f"climate2weather_amd: activation {type(act).__name__} -- the conv epilogues implement SiLU (train.py:171) and "
                                      "ReLU (the default of model/nn.py:118); any other `activation` of model/nn.py:118,156 needs the "
                                      "reference module (INTEGRATION.md, 'What raises').")
        self.in_channels, self.out_channels, self.mod_features, self.spatial = in_channels, out_channels, mod_features, spatial
        self.hidden_channels = list(hidden_channels)
        self.hidden_blocks = list(hidden_blocks)
        self.attention_levels = list(attention_levels)
        conv_kw = dict(kernel_size=3, padding=1)

        def block(c: int) -> ModResidualBlock:
            return ModResidualBlock(
                project=torch.nn.Sequential(torch.nn.Linear(mod_features, c), torch.nn.Unflatten(-1, (-1, 1, 1))),
                residue=torch.nn.Sequential(torch.nn.Identity(), torch.nn.Conv2d(c, c, **conv_kw), activation(),
                                            torch.nn.Conv2d(c, c, **conv_kw)),
            )

        heads, tails, descent, ascent = [], [], [], []
        for i, nblk in enumerate(self.hidden_blocks):  # creation order = the reference's (RNG parity)
            c = self.hidden_channels[i]
            if i > 0:
                cp = self.hidden_channels[i - 1]
                heads.append(torch.nn.Sequential(torch.nn.Conv2d(cp, c, stride=2, **conv_kw)))
                tails.append(torch.nn.Sequential(torch.nn.Identity(), torch.nn.Upsample(scale_factor=(2, 2), mode="nearest"),
                                                 torch.nn.Conv2d(c, cp, **conv_kw)))
            else:
                heads.append(torch.nn.Conv2d(in_channels, c, **conv_kw))
                tails.append(torch.nn.Conv2d(c, out_channels, **conv_kw))
            dl, al = [], []
            for _ in range(nblk):
                dl.append(block(c))
                al.append(block(c))
                if i in self.attention_levels:
                    dl.append(AttentionBlock(c))
                    al.append(AttentionBlock(c))
            descent.append(torch.nn.ModuleList(dl))
            ascent.append(torch.nn.ModuleList(al))
        self.heads = torch.nn.ModuleList(heads)
        self.tails = torch.nn.ModuleList(reversed(tails))
        self.descent = torch.nn.ModuleList(descent)
        self.ascent = torch.nn.ModuleList(reversed(ascent))
        # Conv2d weights take their final memory format NOW ([Cout][kh][kw][Cin] = channels_last, the K-contiguous operand of the
        # implicit GEMM; same values, same OIHW shape).  The engine later only moves their storage into its flat buffer, with
        # these strides: whatever captured the parameters in between -- torch DDP builds its bucket views from the strides it
        # sees at construction and silently permutes gradients if they change -- keeps seeing the same layout.
        with torch.no_grad():
            for p in self.parameters():
                if p.dim() == 4:
                    p.data = p.data.contiguous(memory_format=torch.channels_last)

    # ---- static description consumed by the engine
    def spec(self) -> List[LevelSpec]:
        L = len(self.hidden_blocks)
        levels: List[LevelSpec] = []
        for i in range(L):
            c = self.hidden_channels[i]
            lv = LevelSpec(c, head_key=("heads.0" if i == 0 else f"heads.{i}.0"),
                           tail_key=(f"tails.{L - 1}" if i == 0 else f"tails.{L - 1 - i}.2"))
            per = 2 if i in self.attention_levels else 1
            for bi in range(self.hidden_blocks[i]):
                lv.descent.append(BlockSpec("res", f"descent.{i}.{bi * per}", c))
                lv.ascent.append(BlockSpec("res", f"ascent.{L - 1 - i}.{bi * per}", c))
                if per == 2:
                    lv.descent.append(BlockSpec("attn", f"descent.{i}.{bi * per + 1}", c))
                    lv.ascent.append(BlockSpec("attn", f"ascent.{L - 1 - i}.{bi * per + 1}", c))
            levels.append(lv)
        return levels

    def forward(self, x, y):  # pragma: no cover - not on the product path
        raise RuntimeError("climate2weather_amd.nn.UNet holds parameters only; call ScoreUNet.forward (HIP engine)")
