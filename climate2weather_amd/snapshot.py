"""Network snapshots: import the reference's ``network-snapshot-*.pkl`` and write the same thing.

The reference pickles ``util.EasyDict(dataset_kwargs=..., pipeline=<thor.pipelines.SDAPipeline>, ema=<fp16
model.score.ScoreUNet on the CPU>)`` (``training_loop.py:249-265``) and the sampler driver unpickles it and uses the
module object directly (``exp/downscaling.py:110-126``).  Unpickling that file normally needs the reference's packages
(``model``, ``thor``, ``util``, ``zuko``) on the path.  ``load_network_snapshot`` does not: class references into those
packages are mapped to stand-ins that only hold state, the module tree's ``state_dict`` (the 228 reference key names) is
read off, the constructor arguments are inferred from the tensor shapes, and the weights are loaded into
``climate2weather_amd.score.ScoreUNet``.  Everything outside an allow-list of modules is refused, so the loader does not
execute arbitrary pickled callables.
"""
from __future__ import annotations

import copy
import io
import pickle
import re
from typing import Any, Dict, Optional

import torch

from .pipelines import SDAPipeline
from .score import ScoreUNet
from .util import EasyDict

_REF_PACKAGES = ("model", "zuko", "thor", "src")
_SAFE_PREFIXES = ("torch", "collections", "numpy", "_codecs", "copyreg", "builtins")
_SAFE_BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray", "complex", "slice",
                  "range", "object", "getattr"}


class _RefModule(torch.nn.Module):
    """Stand-in for any ``nn.Module`` subclass of the reference: ``nn.Module.__setstate__`` restores ``_parameters`` /
    ``_modules``, which is all ``state_dict()`` needs."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("reference module stand-in: weights only")


class _RefObject:
    """Stand-in for a plain reference object (keeps its ``__dict__``)."""


class _SnapshotUnpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str) -> Any:
        root = module.split(".")[0]
        if module == "util" and name == "EasyDict":
            return EasyDict
        if module in ("thor.pipelines", "src.thor.pipelines") and name == "SDAPipeline":
            return SDAPipeline
        if module.startswith("climate2weather_amd"):
            return super().find_class(module, name)
        if root in _REF_PACKAGES:
            # modules in model/ and zuko.nn are nn.Modules; anything else from the reference keeps its attributes only
            return _RefModule if root in ("model", "zuko") else _RefObject
        if root == "builtins":
            if name in _SAFE_BUILTINS:
                return super().find_class(module, name)
            raise pickle.UnpicklingError(f"refusing builtins.{name} in a network snapshot")
        if any(module == p or module.startswith(p + ".") for p in _SAFE_PREFIXES):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing {module}.{name} in a network snapshot")


def infer_config(sd: Dict[str, torch.Tensor]) -> Dict[str, Any]:
    """Constructor keywords of ``ScoreUNet`` (``train.py:164-173``) from a reference state_dict."""
    head = sd["unet.heads.0.weight"]
    hidden = [head.shape[0]]
    i = 1
    while f"unet.heads.{i}.0.weight" in sd:
        hidden.append(sd[f"unet.heads.{i}.0.weight"].shape[0])
        i += 1
    blocks, attn = [], []
    for lvl in range(len(hidden)):
        idx = sorted({int(m.group(1)) for k in sd for m in [re.match(rf"unet\.descent\.{lvl}\.(\d+)\.", k)] if m})
        res = [j for j in idx if f"unet.descent.{lvl}.{j}.residue.1.weight" in sd]
        blocks.append(len(res))
        if any(f"unet.descent.{lvl}.{j}.qkv.weight" in sd for j in idx):
            attn.append(lvl)
    return dict(channels=int(head.shape[1]), spatial=head.dim() - 2, embedding_dim=int(sd["map_layer0.weight"].shape[0]),
                hidden_channels=[int(c) for c in hidden], hidden_blocks=blocks, attention_levels=attn, kernel_size=int(head.shape[-1]),
                padding_mode="zeros")


def network_from_state_dict(sd: Dict[str, torch.Tensor], device=None, precision: Optional[str] = None) -> ScoreUNet:
    sd = {k: v for k, v in sd.items() if not k.endswith(".eps")}  # zuko's LayerNorm may carry an `eps` buffer
    net = ScoreUNet(activation=torch.nn.SiLU, **infer_config(sd))
    net.load_state_dict({k: v.to(torch.float32) for k, v in sd.items()})
    if device is not None:
        net = net.to(device)
    if precision is not None:
        net.precision = precision
    return net.eval().requires_grad_(False)


def load_network_snapshot(path_or_file, device=None, precision: Optional[str] = None) -> EasyDict:
    """-> EasyDict(ema=ScoreUNet (this package's, fp32 master weights), pipeline=SDAPipeline, dataset_kwargs=...,
    markov_order=window // 2) from a reference (or own) snapshot file."""
    f = open(path_or_file, "rb") if isinstance(path_or_file, (str, bytes)) or hasattr(path_or_file, "__fspath__") else path_or_file
    try:
        data = _SnapshotUnpickler(f).load()
    finally:
        if f is not path_or_file:
            f.close()
    ema = data["ema"]
    if isinstance(ema, ScoreUNet):
        net = ema.float()
        if device is not None:
            net = net.to(device)
        if precision is not None:
            net.precision = precision
        net = net.eval().requires_grad_(False)
    else:
        net = network_from_state_dict(ema.state_dict(), device=device, precision=precision)
    pipe = data.get("pipeline")
    if not isinstance(pipe, SDAPipeline):
        pipe = SDAPipeline(eta=getattr(pipe, "eta", 1e-3))
    out = EasyDict(ema=net, pipeline=pipe, dataset_kwargs=data.get("dataset_kwargs"))
    try:
        out.markov_order = int(data["dataset_kwargs"]["train"]["window"]) // 2  # exp/downscaling.py:113-114
    except Exception:
        out.markov_order = None
    return out


def save_network_snapshot(path: str, net: ScoreUNet, pipeline: SDAPipeline, dataset_kwargs: Optional[dict] = None) -> str:
    """``training_loop.py:249-265``: deep copy -> CPU -> eval -> no grad -> fp16, pickled next to the pipeline and the
    dataset keywords.  The file unpickles wherever this package is importable (the reference's sampler driver included)."""
    snap = EasyDict(dataset_kwargs=dataset_kwargs, pipeline=pipeline)
    snap.ema = copy.deepcopy(net).cpu().eval().requires_grad_(False).to(torch.float16)
    with open(path, "wb") as f:
        pickle.dump(snap, f)
    return path


def snapshot_bytes(net: ScoreUNet, pipeline: SDAPipeline, dataset_kwargs: Optional[dict] = None) -> bytes:
    buf = io.BytesIO()
    snap = EasyDict(dataset_kwargs=dataset_kwargs, pipeline=pipeline)
    snap.ema = copy.deepcopy(net).cpu().eval().requires_grad_(False).to(torch.float16)
    pickle.dump(snap, buf)
    return buf.getvalue()
