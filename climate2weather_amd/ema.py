"""Drop-in for ``thor.ema.StandardEMA`` (src/thor/ema.py:6-42): the same constructor and the same six entry points
(``update``, ``get``, ``reset``, ``state_dict``, ``load_state_dict`` and the ``emas`` / ``rates`` attributes the training
loop reads, training_loop.py:125,250-265,390).

What is different underneath: every tracked copy of an engine-backed ``ScoreUNet`` is ONE flat fp32 buffer in HBM
(engine.py::Layout), so an update is one fused ``p_ema <- r p_ema + (1 - r) p`` kernel over 72 M elements per rate instead
of 228 ``mul_`` / ``add_`` pairs, and ``reset`` is one device copy.  Copies that are not on the GPU engine (CPU modules,
other module classes) go through the same three per-tensor primitives below.
"""
from __future__ import annotations

import copy
from typing import Iterator, List, Optional, Tuple

import torch

from . import ops


def _flat_buffers(src, dst) -> Optional[Tuple[torch.Tensor, "object"]]:
    """(flat buffer of ``src``, engine of ``dst``) when both modules keep their weights in a flat GPU buffer of the same device."""
    if not (hasattr(src, "_get_engine") and hasattr(dst, "_get_engine")):
        return None
    first = next(iter(src.parameters()), None)
    if first is None or not (first.is_cuda or ops.EMULATED):  # CPU modules: the per-tensor primitives (and no engine is created for them)
        return None
    e_src, e_dst = src._get_engine(), dst._get_engine()
    if e_src.flat is None or e_dst.flat is None or e_src.flat.device != e_dst.flat.device or e_src.flat.numel() != e_dst.flat.numel():
        return None
    return e_src.flat, e_dst


class StandardEMA:
    """Exponential moving averages of a network's parameters, one deep copy per rate."""

    def __init__(self, net, rates=[0.9999]):
        self.net = net
        self.rates = list(rates)
        with torch.no_grad():
            self.emas = [copy.deepcopy(net) for _ in self.rates]

    def _tracked(self) -> Iterator[Tuple[float, torch.nn.Module]]:
        return zip(self.rates, self.emas)

    def _blend(self, avg: torch.nn.Module, rate: float) -> None:
        """avg <- rate * avg + (1 - rate) * net.  rate == 0 is a plain copy (``reset``)."""
        pair = _flat_buffers(self.net, avg)
        if pair is not None:
            live, eng = pair
            if rate == 0.0:
                eng.flat.copy_(live)
            else:
                ops.ema_update(eng.flat, live, live.numel(), float(rate))
            eng.weights_changed()
            return
        for src, dst in zip(self.net.parameters(), avg.parameters()):  # callers hold torch.no_grad()
            if rate == 0.0:
                dst.copy_(src)
            else:
                dst.mul_(rate).add_(src, alpha=1.0 - rate)

    @torch.no_grad()
    def update(self, **kwargs) -> None:
        """One averaging step per rate (src/thor/ema.py:23-27; keyword arguments such as ``cur_ndata`` are accepted and unused)."""
        for rate, avg in self._tracked():
            self._blend(avg, float(rate))

    @torch.no_grad()
    def reset(self) -> None:
        """Every copy takes the live network's current parameters."""
        for _, avg in self._tracked():
            self._blend(avg, 0.0)

    @torch.no_grad()
    def get(self) -> List[Tuple[torch.nn.Module, str]]:
        """[(copy, "-<rate>")] with buffers synchronised from the live network first -- the suffix is what the snapshot file
        names carry (training_loop.py:250-252)."""
        for _, avg in self._tracked():
            for src, dst in zip(self.net.buffers(), avg.buffers()):
                dst.copy_(src)
        return [(avg, "-%.6f" % rate) for rate, avg in self._tracked()]

    def state_dict(self) -> dict:
        return {"rates": self.rates, "emas": [avg.state_dict() for avg in self.emas]}

    def load_state_dict(self, state: dict) -> None:
        self.rates = state["rates"]
        for avg, saved in zip(self.emas, state["emas"]):
            avg.load_state_dict(saved)
