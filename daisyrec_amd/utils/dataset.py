"""Loader boundary (reference: daisy/utils/dataset.py:5-27).

``BasicDataset`` / ``get_dataloader`` keep the reference's names and behaviour so
callers (run_examples/test.py:93-94) are unchanged; the native ``MF.fit`` never
iterates the DataLoader — it reads ``dataset.data`` (the int32 [N,3] triples) once,
moves it to HBM and replays the loader's index order on the device.

``DeviceTripleLoader`` is the throughput-mode loader: triples stay resident in HBM
and each epoch's order is a device-side Philox permutation.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .. import ops


def get_dataloader(ds, batch_size, shuffle, num_workers=4):
    """dataset.py:5-7."""
    return DataLoader(ds, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers)


class BasicDataset(Dataset):
    """dataset.py:10-27: array-like <u, i, j> samples."""

    def __init__(self, samples):
        super().__init__()
        self.data = samples

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        row = self.data[index]
        return row[0], row[1], row[2]


class CandidatesDataset(Dataset):
    """dataset.py:29-38 (used by MF.rank's caller, test.py:118)."""

    def __init__(self, ucands):
        super().__init__()
        self.data = ucands

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        return torch.tensor(self.data[index][0]), torch.tensor(self.data[index][1])


class DeviceTripleLoader:
    """HBM-resident triples + per-epoch device permutation (shuffle=True semantics:
    every epoch is a fresh uniform permutation; last batch partial)."""

    def __init__(self, triples, batch_size, shuffle=True, seed=2022, device="cuda"):
        t = torch.as_tensor(np.asarray(triples) if not isinstance(triples, torch.Tensor) else triples)
        self.triples = t.to(torch.int32).contiguous().to(device)
        self.batch_size = int(batch_size)
        self.shuffle = bool(shuffle)
        self.seed = int(seed)
        self.epoch = 0

    def __len__(self):
        return (self.triples.shape[0] + self.batch_size - 1) // self.batch_size

    def next_epoch_order(self):
        """int64 device permutation for the next epoch (None when not shuffling)."""
        e = self.epoch
        self.epoch += 1
        if not self.shuffle:
            return None
        return ops.randperm(self.triples.shape[0], self.seed, e, device=self.triples.device)
