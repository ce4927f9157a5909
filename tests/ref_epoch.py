"""The reference's epoch control flow (trainer.py:177-241, train.py:75-124) driven by the oracle -- test infrastructure.

Used by tests/test_api_gpu.py (tiny model, compared epoch by epoch) and by tests/golden/make_epoch_default.py (the
default-config N = 1000 fixture of SURVEY.md section 8(d): "ELBO curve matching the reference at epoch 1").  Adam
state is carried across epochs, NumPy's global legacy stream is consumed in the reference's order: batch shuffle
(mlx_data/dataloader.py:94), one coin per decoder time step (models/decoder.py:180, also at ratio 0.0), the logging
forward after batch 0 and every 25th batch (trainer.py:336-363, Q16), the 20-batch "true" train loss and the
validation pass at teacher forcing 0 (trainer.py:116-175, 418-487), the 64-row latent statistics pass and the
monitoring MI with log(mean_var + 1e-8) (trainer.py:524-575, Q20).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch

import arcvae_oracle as O


def synthetic_json(n: int, vocab: int, max_length: int, seed: int = 67) -> dict:
    """SELFIES-shaped stand-in with the reference's JSON schema (train.py:79-83,102): random body tokens in [3, V),
    EOS = 2, lengths in [20, max_length - 1) (short sequences for tiny max_length), TPSA-like gamma properties."""
    rs = np.random.RandomState(seed)
    lo = min(20, max(2, max_length // 3))
    seqs, mols = [], []
    for _ in range(n):
        ln = int(rs.randint(lo, max_length - 1))
        seqs.append([int(t) for t in rs.randint(3, vocab, size=ln)] + [2])
        mols.append({"tpsa": float(rs.gamma(4.0, 20.0))})
    return {"molecules": mols, "tokenized_sequences": seqs, "max_length": max_length}


def split_80_10_10(data: dict, seed: int = 67):
    """train.py:75,86-96: np.random.seed(67), shuffle, 80/10/10 -> index arrays."""
    np.random.seed(seed)
    idx = np.arange(len(data["tokenized_sequences"]))
    np.random.shuffle(idx)
    n = len(idx)
    n_tr, n_va = int(0.8 * n), int(0.1 * n)
    return idx[:n_tr], idx[n_tr:n_tr + n_va], idx[n_tr + n_va:]


def tensorise(data: dict, idx, mean=None, std=None) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """mlx_data/dataloader.py:46-83: pad / truncate to max_length with 0, z-score the properties (train statistics)."""
    T = data["max_length"]
    props = np.array([[data["molecules"][i]["tpsa"]] for i in idx], dtype=np.float32)
    if mean is None:
        mean, std = props.mean(axis=0, keepdims=True), props.std(axis=0, keepdims=True)
        std = np.where(std < 1e-8, 1.0, std)
    toks = np.zeros((len(idx), T), dtype=np.int64)
    for r, i in enumerate(idx):
        s = data["tokenized_sequences"][i][:T]
        toks[r, :len(s)] = s
    return toks, ((props - mean) / std).astype(np.float32), mean, std


def reference_epoch(cfg: O.Config, p: Dict[str, np.ndarray], m, v, train, val, bs: int, T: int, lr: float, epoch: int,
                    total_epochs: int, hp: dict, dtype=torch.float32) -> Dict[str, float]:
    """One epoch; updates p, m, v in place and returns the metrics train.py:219-233 appends to the history."""
    beta = O.compute_beta(epoch, hp["beta_start"], hp["beta_end"], hp["warmup"])
    tf = O.compute_teacher_forcing_ratio(epoch, total_epochs)
    hy = dict(beta=beta, lambda_collapse=hp["lambda_collapse"], free_bits=hp["free_bits"], lambda_mi=hp["lambda_mi"])
    eps0 = lambda n: np.zeros((n, cfg.Z), np.float32)   # no loss value depends on eps (Q2)

    def fwd(xb, cb, ratio):
        coins = O.draw_coins(np.random, T, ratio)
        with torch.no_grad():
            return O.complete_vae_loss(O.to_torch(p, dtype), cfg, torch.tensor(xb), torch.tensor(cb, dtype=dtype),
                                       torch.tensor(eps0(len(xb)), dtype=dtype), coins, **hy)

    idx = np.arange(len(train[0]))
    np.random.shuffle(idx)
    for bi, i in enumerate(range(0, len(idx), bs)):
        sel = idx[i:i + bs]
        coins = O.draw_coins(np.random, T, tf)
        O.train_step(p, m, v, cfg, train[0][sel], train[1][sel], eps0(len(sel)), coins, lr, dtype=dtype, **hy)
        if bi == 0 or bi % 25 == 0:
            fwd(train[0][sel], train[1][sel], tf)

    def evaluate(data, limit):
        tot, n = np.zeros(4), 0
        for bi, i in enumerate(range(0, len(data[0]), bs)):
            if limit is not None and bi >= limit:
                break
            d = fwd(data[0][i:i + bs], data[1][i:i + bs], 0.0)
            tot += [float(d["total_loss"]), float(d["recon_loss"]), float(d["kl_loss"]), float(d["collapse_penalty"])]
            n += 1
        return tot / max(n, 1)

    tr = evaluate(train, 20)
    va = evaluate(val, None)
    with torch.no_grad():                                # trainer.py:524-575: first 64 training rows, MI with +1e-8
        pe = {k[len("encoder."):]: t for k, t in O.to_torch(p, dtype).items() if k.startswith("encoder.")}
        mu, lv = O.encoder_forward(pe, torch.tensor(train[0][:64]), torch.tensor(train[1][:64], dtype=dtype), cfg.L)
        mi = float(O.mutual_information(mu, lv, log_eps=1e-8))
    return dict(train_loss=tr[0], train_recon=tr[1], train_kl=tr[2], train_collapse=tr[3], val_loss=va[0],
                val_recon=va[1], val_kl=va[2], val_collapse=va[3], beta=beta, teacher_forcing=tf, mutual_info=mi)
