"""Data feed of the training loop (dataset.py:11-126) without h5py: the rank-strided infinite sampler, a seeded
synthetic stand-in for ``COSMODataset`` that yields the same item layout, and a device-resident feed that removes
the DataLoader / host->device path."""
from __future__ import annotations

from typing import Iterator, Optional

import os

import numpy as np
import torch


class InfiniteSampler(torch.utils.data.Sampler):
    """The index stream of dataset.py:11-40 -- item j of a rank is ``perm[e][p % N]`` with ``p = start_idx + rank + j * num_replicas``,
    ``e = p // N`` and ``perm[e]`` the permutation ``RandomState(hash((seed, e)) % 2**31).shuffle(arange(N))`` -- kept as a CURSOR
    instead of a generator: ``take(n)`` returns the next n indices as one array (a batch is one vectorised gather per epoch it
    touches, not n Python iterations), ``positions(n)`` the raw (epoch, offset) arithmetic for a device-side gather
    (DeviceWindowFeed), and iteration (the torch Sampler protocol: DataLoader, ``next(iter(...))``) draws from the same cursor in
    blocks.  The sequence is a bit-exact contract with the reference (tests/golden/kat.json).  (The reference's ``__init__`` calls
    ``Sampler.__init__(dataset)``, which torch >= 2.2 rejects.)"""

    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, start_idx=0):
        n = len(dataset)
        if n <= 0 or num_replicas <= 0 or not 0 <= rank < num_replicas:
            raise AssertionError("InfiniteSampler: empty dataset or rank outside [0, num_replicas)")
        self.dataset_size, self.stride, self.shuffle, self.seed = n, num_replicas, shuffle, seed
        self.start_idx = start_idx + rank
        self.cursor = self.start_idx  # position p of the next item
        self._perm = {}  # epoch -> permutation (the two most recent are kept)

    def permutation(self, epoch: int) -> np.ndarray:
        perm = self._perm.get(epoch)
        if perm is None:
            perm = np.arange(self.dataset_size)
            if self.shuffle:
                np.random.RandomState(hash((self.seed, epoch)) % (1 << 31)).shuffle(perm)
            for old in [e for e in self._perm if e < epoch - 1]:
                del self._perm[old]
            self._perm[epoch] = perm
        return perm

    def _runs(self, p: int, n: int):
        """n items from position p on: ([(epoch, first offset inside the epoch, count)] runs of constant epoch, offsets stepping by
        ``self.stride`` inside a run; position after them)."""
        runs, left = [], n
        while left > 0:
            e, off = divmod(p, self.dataset_size)
            cnt = min(left, (self.dataset_size - off + self.stride - 1) // self.stride)
            runs.append((e, off, cnt))
            p += cnt * self.stride
            left -= cnt
        return runs, p

    def positions(self, n: int):
        """Advance the cursor by n items and return their runs (see _runs)."""
        runs, self.cursor = self._runs(self.cursor, n)
        return runs

    def _gather(self, runs) -> np.ndarray:
        parts = [self.permutation(e)[off: off + cnt * self.stride: self.stride] for e, off, cnt in runs]
        return parts[0] if len(parts) == 1 else np.concatenate(parts)

    def take(self, n: int) -> np.ndarray:
        return self._gather(self.positions(n))

    def __iter__(self) -> Iterator[int]:
        """Every iterator walks the stream from ``start_idx`` on its own position, like the reference's generator (dataset.py:28-40);
        the object's cursor (take / positions) is not touched."""
        p = self.start_idx
        while True:
            runs, p = self._runs(p, 256)
            for i in self._gather(runs).tolist():
                yield i


class SyntheticWindowDataset(torch.utils.data.Dataset):
    """Same item contract as ``COSMODataset`` (dataset.py:60-126): an array x[N, F, H, W] and items
    ``x[i : i + window].reshape(window * F, H, W)`` (channel = tau*F + c), over synthetic fields
    ``0.5 * randn(seed) + 0.5`` (quantile-normalised COSMO is ~[0, 1])."""

    def __init__(self, n_frames: int, n_vars: int, height: int, width: int, window: int, seed: int = 0, flatten: bool = True):
        g = torch.Generator().manual_seed(seed)
        self.data = torch.randn(n_frames, n_vars, height, width, generator=g) * 0.5 + 0.5
        self._window, self._flatten = window, flatten

    @property
    def window(self):
        return self._window

    @property
    def flatten(self):
        return self._flatten

    def __len__(self):
        return self.data.shape[0] - self._window + 1

    def load_window(self, i: int):
        return self.data[i : i + self._window]

    def __getitem__(self, i: int):
        x = self.load_window(i)
        return x.reshape(-1, *x.shape[2:]) if self._flatten else x


class COSMODataset(torch.utils.data.Dataset):
    """``dataset.COSMODataset`` (dataset.py:60-126) with the same constructor keywords, properties and item contract, over
    an array ``x[N, F, H, W]`` that is read ONCE into memory (the reference's ``cached=True``) so that ``DeviceWindowFeed``
    can keep it in HBM.  ``data_path``: the reference's ``.h5`` file with dataset ``"x"`` (needs ``h5py``, which the
    reference also needs), a ``.npy`` file, or an in-memory array / tensor."""

    def __init__(self, data_path, num_features: int, spatial_res: int, cached: bool = True, window: int = 13, flatten: bool = True):
        self._window, self._flatten = window, flatten
        self._data_path = data_path if isinstance(data_path, str) else "<array>"
        if isinstance(data_path, str):
            path = os.path.abspath(data_path)
            assert os.path.isfile(path), path
            ext = os.path.splitext(path)[-1]
            if ext == ".h5":
                try:
                    import h5py
                except ImportError as e:  # pragma: no cover
                    raise ImportError("reading the reference's .h5 files needs h5py; convert to .npy or pass the array") from e
                with h5py.File(path, mode="r") as f:
                    arr = f["x"][:]  # dataset.py:73,81-84
            elif ext == ".npy":
                arr = np.load(path)
            else:
                raise ValueError(f"unsupported dataset file {path}")
            self.data = torch.from_numpy(np.ascontiguousarray(arr)).float()
        else:
            self.data = torch.as_tensor(data_path).float()
        assert self.data.dim() == 4, self.data.shape
        assert self.data.shape[-1] == self.data.shape[-2] == spatial_res  # dataset.py:90
        self.spatial_res = spatial_res
        assert num_features == self.num_features, (
            f"The number of specified features ({num_features}) does not match the number of features in the data ({self.num_features}).")

    window = property(lambda self: self._window)
    flatten = property(lambda self: self._flatten)
    raw_data_shape = property(lambda self: tuple(self.data.shape))
    raw_spatial_res = property(lambda self: self.spatial_res)
    num_features = property(lambda self: self.data.shape[-3])
    data_path = property(lambda self: self._data_path)

    def __len__(self) -> int:
        return self.data.shape[0] - self._window + 1

    def load_window(self, i: int):
        return self.data[i : i + self._window]

    def __getitem__(self, i: int):
        x = self.load_window(i)
        return x.flatten(0, 1) if self._flatten else x


class WindowBatch:
    """A training batch that has not been gathered: ``data`` (N, F, H, W) fp32 on the device and the first frame of each of the B
    windows.  Window b is the w * F * H * W contiguous floats from frame ``first[b]`` on (dataset.py:114-126 reshapes exactly that
    block), so the trainer's input conversion reads it in place (ops.windows_to_nhwc_noise) and the (B, w*F, H, W) tensor is never
    written.  ``materialize()`` is the plain tensor for every other consumer."""

    def __init__(self, data: torch.Tensor, first: torch.Tensor, window: int):
        self.data, self.first, self.window = data, first, window
        n, f, h, w = data.shape
        self.shape = torch.Size((first.numel(), window * f, h, w))
        self.device, self.dtype, self.is_cuda = data.device, data.dtype, data.is_cuda
        self._ar = None

    def offsets(self) -> torch.Tensor:
        """float offset of every window inside ``data`` (int64, on the device)"""
        return self.first * (self.data.stride(0))

    def materialize(self) -> torch.Tensor:
        ar = torch.arange(self.window, device=self.data.device)
        rows = (self.first[:, None] + ar[None, :]).reshape(-1)
        return torch.index_select(self.data, 0, rows).view(tuple(self.shape))


class DeviceWindowFeed:
    """Whole (normalised) array resident in HBM; a batch is an index gather on the device -- no DataLoader workers, no
    pinned staging, no H2D copy per step.  8 y x 8760 h x 4 x 128^2 fp32 = 18 GB fits MI355X's 288 GB many times."""

    def __init__(self, dataset, device, rank=0, num_replicas=1, seed=0, start_idx=0, shuffle=True):
        self.data = dataset.data.to(device)
        self.window = dataset.window
        self.sampler = InfiniteSampler(dataset, rank, num_replicas, shuffle, seed, start_idx)
        self._ar = torch.arange(self.window, device=device)
        self._perm_dev = {}  # epoch -> that epoch's permutation in HBM (uploaded once per epoch, not once per batch)
        self._steps = {}  # batch size -> arange(batch) * stride on the device

    def _indices(self, batch: int) -> torch.Tensor:
        """First frame of each of the next ``batch`` windows, int64 on the device.  The index stream lives on the device too: an
        epoch's permutation is uploaded once, a batch is ``perm[off + arange(batch) * stride]`` -- one small gather, no host list, no
        pinned staging, no host->device copy on the step's critical path."""
        dev = self.data.device
        runs = self.sampler.positions(batch)
        parts = []
        for e, off, cnt in runs:
            perm = self._perm_dev.get(e)
            if perm is None:
                for old in [k for k in self._perm_dev if k < e - 1]:
                    del self._perm_dev[old]
                perm = self._perm_dev[e] = torch.from_numpy(self.sampler.permutation(e)).to(dev)
            steps = self._steps.get(cnt)
            if steps is None:
                steps = self._steps[cnt] = torch.arange(cnt, device=dev, dtype=torch.int64) * self.sampler.stride
            parts.append(perm[steps + off])
        return parts[0] if len(parts) == 1 else torch.cat(parts)

    def next_batch(self, batch: int, lazy: bool = False):
        """``lazy``: return a WindowBatch (indices only) instead of the gathered (B, w*F, H, W) tensor; Trainer.step takes either."""
        idx = self._indices(batch)
        if lazy:
            return WindowBatch(self.data, idx, self.window)
        # (B, w*F, H, W): frame rows idx[b] .. idx[b] + w - 1 by ONE index_select over the frame axis (advanced indexing of the 4-D
        # array with a (B, w) index costs 0.4 ms of host time per call -- on the critical path of a loop that synchronises every step)
        n, f, h, w_ = self.data.shape
        rows = (idx[:, None] + self._ar[None, :]).reshape(-1)
        return torch.index_select(self.data, 0, rows).view(batch, self.window * f, h, w_)
