"""Network snapshots: import the reference's ``network-snapshot-*.pkl`` and write the same thing.

The reference pickles ``util.EasyDict(dataset_kwargs=..., pipeline=<thor.pipelines.SDAPipeline>, ema=<fp16
model.score.ScoreUNet on the CPU>)`` (``training_loop.py:249-265``) and the sampler driver unpickles it and uses the
module object directly (``exp/downscaling.py:110-126``).  Unpickling that file normally needs the reference's packages
(``model``, ``thor``, ``util``, ``zuko``) on the path.  ``load_network_snapshot`` does not: class references into those
packages are mapped to stand-ins that only hold state, the module tree's ``state_dict`` (the 228 reference key names) is
read off, the constructor arguments are inferred from the tensor shapes, and the weights are loaded into
``climate2weather_amd.score.ScoreUNet``.  Globals are resolved from an allow-list of exact (module, name) pairs (tensor
rebuild helpers, ``torch.nn`` module classes, storages through the weights-only loader, this package's own classes); dotted
names and everything else are refused.
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

_REF_MODULE_PACKAGES = ("model", "zuko")   # nn.Module subclasses of the reference: only their _parameters / _modules are read
_REF_OBJECT_PACKAGES = ("thor", "src")     # plain objects of the reference: attributes only

# Exact (module, name) pairs a snapshot may reference; nothing is matched by prefix and no dotted name is resolved (pickle protocol 4
# resolves "a.b" through getattr, which would reach e.g. torch.serialization.os.system through an allowed module).
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_parameter_with_state"),
    ("torch", "Size"), ("torch", "device"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
    ("_codecs", "encode"),
    ("climate2weather_amd.score", "ScoreUNet"), ("climate2weather_amd.nn", "UNet"), ("climate2weather_amd.nn", "ModResidualBlock"),
    ("climate2weather_amd.nn", "AttentionBlock"), ("climate2weather_amd.pipelines", "SDAPipeline"), ("climate2weather_amd.util", "EasyDict"),
}
_SAFE_BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray", "complex", "slice", "range"}
_TORCH_DTYPES = {n for n in dir(torch) if isinstance(getattr(torch, n), torch.dtype)}
_TORCH_NN_MODULES = ("torch.nn.modules.activation", "torch.nn.modules.container", "torch.nn.modules.conv", "torch.nn.modules.flatten",
                     "torch.nn.modules.linear", "torch.nn.modules.upsampling", "torch.nn.modules.normalization", "torch.nn.modules.dropout",
                     "torch.nn.modules.pooling")


def _storage_from_bytes(b: bytes):
    """``torch.storage._load_from_bytes`` re-enters ``torch.load(weights_only=False)`` = a second, unrestricted unpickler; storages
    need no more than the weights-only loader."""
    return torch.load(io.BytesIO(b), weights_only=True)


class _RefModule(torch.nn.Module):
    """Stand-in for any ``nn.Module`` subclass of the reference: ``nn.Module.__setstate__`` restores ``_parameters`` /
    ``_modules``, which is all ``state_dict()`` needs."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("reference module stand-in: weights only")


class _RefObject:
    """Stand-in for a plain reference object (keeps its ``__dict__``)."""


class _SnapshotUnpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str) -> Any:
        if "." in name:
            raise pickle.UnpicklingError(f"refusing dotted global {module}.{name} in a network snapshot")
        root = module.split(".")[0]
        if module == "util" and name == "EasyDict":
            return EasyDict
        if module in ("thor.pipelines", "src.thor.pipelines") and name == "SDAPipeline":
            return SDAPipeline
        if root in _REF_MODULE_PACKAGES:
            return _RefModule
        if root in _REF_OBJECT_PACKAGES:
            return _RefObject
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            return _storage_from_bytes
        if module == "builtins" and name in _SAFE_BUILTINS:
            return super().find_class(module, name)
        if module == "torch" and (name in _TORCH_DTYPES or name.endswith("Storage")):
            obj = super().find_class(module, name)
            if isinstance(obj, torch.dtype) or (isinstance(obj, type) and name.endswith("Storage")):
                return obj
        if module in _TORCH_NN_MODULES:
            obj = super().find_class(module, name)
            if isinstance(obj, type) and issubclass(obj, torch.nn.Module):
                return obj
        if (module, name) in _SAFE_GLOBALS:
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


def network_from_state_dict(sd: Dict[str, torch.Tensor], device=None, precision: Optional[str] = None, activation=torch.nn.SiLU) -> ScoreUNet:
    """``activation``: the residual blocks' activation class -- not recorded in a state_dict (train.py:171 passes SiLU, the UNet's own
    default is ReLU); ``load_network_snapshot`` reads it off the pickled module tree."""
    sd = {k: v for k, v in sd.items() if not k.endswith(".eps")}  # zuko's LayerNorm may carry an `eps` buffer
    net = ScoreUNet(activation=activation, **infer_config(sd))
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
        kinds = {type(m) for m in ema.modules()}
        act = torch.nn.ReLU if torch.nn.ReLU in kinds and torch.nn.SiLU not in kinds else torch.nn.SiLU
        net = network_from_state_dict(ema.state_dict(), device=device, precision=precision, activation=act)
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
