"""The slice of the reference's ``util.py`` the hot path touches: seeding (util.py:27-29) and the name-based
construction seam (util.py:56-127) through which ``train.py`` selects the network / pipeline / EMA / lr classes."""
from __future__ import annotations

import importlib
import random
from typing import Any

import numpy as np
import torch


def set_random_seed(*args) -> int:
    """util.py:27-29: seed = hash(args) % 2**31 (CPython int-tuple hashes are unsalted), applied to python, numpy, torch."""
    seed = hash(args) % (1 << 31)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed


class EasyDict(dict):
    """Attribute-style dict (util.py:36-49)."""

    def __getattr__(self, name: str) -> Any:
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name: str, value: Any) -> None:
        self[name] = value

    def __delattr__(self, name: str) -> None:
        del self[name]


def get_obj_by_name(name: str) -> Any:
    parts = name.split(".")
    for cut in range(len(parts), 0, -1):
        try:
            obj = importlib.import_module(".".join(parts[:cut]))
        except ImportError:
            continue
        for p in parts[cut:]:
            obj = getattr(obj, p)
        return obj
    raise ImportError(name)


def construct_class_by_name(*args, class_name: str, **kwargs) -> Any:
    """util.py:125-127: ``class_name="climate2weather_amd.score.ScoreUNet"`` is the whole drop-in."""
    return get_obj_by_name(class_name)(*args, **kwargs)


def call_func_by_name(*args, func_name: str, **kwargs) -> Any:
    return get_obj_by_name(func_name)(*args, **kwargs)
