"""Data feed of the training loop (dataset.py:11-126) without h5py: the rank-strided infinite sampler, a seeded
synthetic stand-in for ``COSMODataset`` that yields the same item layout, and a device-resident feed that removes
the DataLoader / host->device path."""
from __future__ import annotations

from typing import Iterator, Optional

import os

import numpy as np
import torch


class InfiniteSampler(torch.utils.data.Sampler):
    """dataset.py:11-40: walks ``order[(start_idx + rank + j*num_replicas) % N]`` over per-epoch permutations seeded with
    ``hash((seed, epoch)) % 2**31``.  (The reference calls ``Sampler.__init__(dataset)``, which torch >= 2.2 rejects.)"""

    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, start_idx=0):
        assert len(dataset) > 0
        assert num_replicas > 0
        assert 0 <= rank < num_replicas
        self.dataset_size = len(dataset)
        self.start_idx = start_idx + rank
        self.stride = num_replicas
        self.shuffle = shuffle
        self.seed = seed

    def __iter__(self) -> Iterator[int]:
        idx = self.start_idx
        epoch = None
        order = None
        while True:
            if epoch != idx // self.dataset_size:
                epoch = idx // self.dataset_size
                order = np.arange(self.dataset_size)
                if self.shuffle:
                    np.random.RandomState(hash((self.seed, epoch)) % (1 << 31)).shuffle(order)
            yield int(order[idx % self.dataset_size])
            idx += self.stride


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
        return self.data[(self.first[:, None] + ar[None, :])].flatten(1, 2)


class DeviceWindowFeed:
    """Whole (normalised) array resident in HBM; a batch is an index gather on the device -- no DataLoader workers, no
    pinned staging, no H2D copy per step.  8 y x 8760 h x 4 x 128^2 fp32 = 18 GB fits MI355X's 288 GB many times."""

    def __init__(self, dataset, device, rank=0, num_replicas=1, seed=0, start_idx=0, shuffle=True):
        self.data = dataset.data.to(device)
        self.window = dataset.window
        self.sampler = iter(InfiniteSampler(dataset, rank, num_replicas, shuffle, seed, start_idx))
        self._ar = torch.arange(self.window, device=device)

    def next_batch(self, batch: int, lazy: bool = False):
        """``lazy``: return a WindowBatch (indices only) instead of the gathered (B, w*F, H, W) tensor; Trainer.step takes either."""
        idx = torch.tensor([next(self.sampler) for _ in range(batch)], dtype=torch.int64)
        if self.data.is_cuda:
            # pinned + asynchronous: a pageable host -> device copy blocks the host until the stream reaches it, i.e. until the PREVIOUS
            # training step has finished on the GPU -- the launch queue then runs dry once per step (0.5 ms of 50.6, rocprof trace).
            # The pinned block belongs to torch's caching host allocator, which does not hand it out again before the copy has run.
            idx = idx.pin_memory().to(self.data.device, non_blocking=True)
        else:
            idx = idx.to(self.data.device)
        if lazy:
            return WindowBatch(self.data, idx, self.window)
        frames = self.data[(idx[:, None] + self._ar[None, :])]  # (B, w, F, H, W)
        return frames.flatten(1, 2)
