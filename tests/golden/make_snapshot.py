#!/usr/bin/env python3
"""Generate tests/golden/ref_snapshot_tiny.pkl: a `network-snapshot-*.pkl` exactly as the reference writes it
(training_loop.py:249-265: pickle of util.EasyDict(dataset_kwargs=..., pipeline=<thor.pipelines.SDAPipeline>, ema=<fp16
model.score.ScoreUNet on the CPU>)), by IMPORTING the reference in the build container.  The fixture is serialized
objects (class paths + tensors), not source.  `util.py` is not importable here (lightning missing): its EasyDict is a
plain dict subclass (util.py:36-49), so a same-named class in a synthesized module `util` yields the same pickle stream.

    python tests/golden/make_snapshot.py
"""
import copy
import importlib.util
import os
import pickle
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [os.path.join(REPO, "oracle", "_shim"), REF]

from model.score import ScoreUNet  # noqa: E402  (reference)

thor = types.ModuleType("thor")
thor.__path__ = []
sys.modules["thor"] = thor
spec = importlib.util.spec_from_file_location("thor.pipelines", f"{REF}/src/thor/pipelines.py")
pipelines = importlib.util.module_from_spec(spec)
sys.modules["thor.pipelines"] = pipelines
spec.loader.exec_module(pipelines)

util = types.ModuleType("util")


class EasyDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


EasyDict.__module__ = "util"
EasyDict.__qualname__ = "EasyDict"
util.EasyDict = EasyDict
sys.modules["util"] = util

TINY = dict(embedding_dim=64, hidden_channels=[32, 64], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3, padding_mode="zeros")
torch.manual_seed(3)
net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **TINY)
snap = EasyDict(dataset_kwargs=dict(train=dict(class_name="dataset.COSMODataset", path="train.h5", window=3, flatten=True),
                                    validation=dict(class_name="dataset.COSMODataset", path="valid.h5", window=3, flatten=True)),
                pipeline=pipelines.SDAPipeline())
snap.ema = copy.deepcopy(net).cpu().eval().requires_grad_(False).to(torch.float16)
with open(os.path.join(HERE, "ref_snapshot_tiny.pkl"), "wb") as f:
    pickle.dump(snap, f)
print("wrote", os.path.getsize(os.path.join(HERE, "ref_snapshot_tiny.pkl")), "bytes")
