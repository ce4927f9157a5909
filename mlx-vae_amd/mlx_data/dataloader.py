"""MoleculeDataset on MI355X (reference mlx_data/dataloader.py:4-111).

Same constructor, attributes (`properties_mean`, `properties_std`, `properties_normalized`) and
`to_batches(batch_size, shuffle)` contract, but the sequences are padded/truncated ONCE into a
device-resident [N,T] int32 matrix (+ [N,C] float32 normalised properties); a batch is one device
gather by the host-shuffled index vector.  The shuffle uses np.random.shuffle on NumPy's global
stream exactly as the reference does (dataloader.py:94), and the final partial batch is yielded
(no drop-last, Q14)."""
from __future__ import annotations

import numpy as np
import torch

from arcvae_hip.module import resolve_device


class MoleculeDataset:
    def __init__(self, tokenized_molecules: list, properties: np.ndarray, max_length: int = 120,
                 pad_token: int = 0, properties_mean: np.ndarray = None, properties_std: np.ndarray = None,
                 device=None):
        self.molecules = tokenized_molecules
        self.max_length = max_length
        self.pad_token = pad_token
        self.properties = np.array(properties, dtype=np.float32)
        if properties_mean is not None and properties_std is not None:
            self.properties_mean = np.array(properties_mean, dtype=np.float32)
            self.properties_std = np.array(properties_std, dtype=np.float32)
        else:
            self.properties_mean = self.properties.mean(axis=0, keepdims=True)
            self.properties_std = self.properties.std(axis=0, keepdims=True)
        if self.properties_mean.ndim == 1:
            self.properties_mean = self.properties_mean[np.newaxis, :]
        if self.properties_std.ndim == 1:
            self.properties_std = self.properties_std[np.newaxis, :]
        self.properties_std = np.where(self.properties_std < 1e-8, 1.0, self.properties_std)
        self.properties_normalized = (self.properties - self.properties_mean) / self.properties_std
        # one-time tensorisation
        toks = np.full((len(tokenized_molecules), max_length), pad_token, dtype=np.int32)
        for i, mol in enumerate(tokenized_molecules):
            n = min(len(mol), max_length)
            toks[i, :n] = np.asarray(mol[:n], dtype=np.int32)
        self.device = resolve_device(device)
        self._tokens = torch.from_numpy(toks).to(self.device)
        self._props = torch.from_numpy(self.properties_normalized.astype(np.float32)).to(self.device)

    def __len__(self) -> int:
        return len(self.molecules)

    def __getitem__(self, idx: int) -> dict:
        return {"molecule": self._tokens[idx], "properties": self._props[idx]}

    def to_batches(self, batch_size: int, shuffle: bool = True):
        indices = np.arange(len(self))
        if shuffle:
            np.random.shuffle(indices)
        idx_dev = torch.from_numpy(indices).to(self.device)
        for i in range(0, len(self), batch_size):
            sel = idx_dev[i:i + batch_size]
            yield self._tokens.index_select(0, sel), self._props.index_select(0, sel)
