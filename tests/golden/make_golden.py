#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the oracle (oracle/arcvae_oracle.py).

The reference itself cannot produce vectors here (its runtime, `mlx`, is not installable offline
and it holds no fixtures of its own: SURVEY.md section 8c), so these are outputs of the CPU
restatement evaluated in float64 ("truth") on seeded inputs -- PARITY UNPINNED against a real MLX
run.  Inputs are regenerated from the seeds below by the tests; only expected outputs are stored.

    python tests/golden/make_golden.py          # rewrites tiny_step.npz, small_step.npz, default_digest.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import arcvae_oracle as O  # noqa: E402
from helpers import DEFAULT, HYPER, SMALL, TINY, make_case  # noqa: E402

SCALARS = ("total_loss", "recon_loss", "kl_loss", "weighted_kl", "collapse_penalty", "mutual_info", "mi_penalty")


def full_case(name, cfg, B, T, tf):
    params, x, cond, eps, coins = make_case(cfg, B, T, tf)
    vals, grads = O.loss_and_grads(params, cfg, x, cond, eps, coins, dtype=torch.float64, **HYPER)
    out = {f"val.{k}": np.asarray(vals[k], dtype=np.float64) for k in SCALARS}
    for k in ("mu", "logvar", "z", "logits"):
        out[f"val.{k}"] = vals[k].astype(np.float32)
    out["val.fed_tokens"] = vals["fed_tokens"].astype(np.int32)
    for k, g in grads.items():
        out[f"grad.{k}"] = g.astype(np.float32)
    # one Adam step (fp32 oracle arithmetic, as the reference) from zero state
    p32 = {k: v.copy() for k, v in params.items()}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v = {k: np.zeros_like(vv) for k, vv in params.items()}
    O.train_step(p32, m, v, cfg, x, cond, eps, coins, 2e-4, **HYPER)
    for k in ("encoder.fc_mu.weight", "encoder.lstm_layer_0.Wh", "decoder.fc_out.weight", "decoder.lstm_layer_1.Wx"):
        out[f"adam1.{k}"] = p32[k]
    out["meta"] = np.array([B, T, int(round(tf * 1000))], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, {k: float(vals[k]) for k in SCALARS[:3]})


def digest_case(name, cfg, B, T, tf):
    """Default shape: scalars, small tensors and checksums of the big ones (keeps the fixture small)."""
    params, x, cond, eps, coins = make_case(cfg, B, T, tf)
    vals, grads = O.loss_and_grads(params, cfg, x, cond, eps, coins, dtype=torch.float64, **HYPER)
    out = {f"val.{k}": np.asarray(vals[k], dtype=np.float64) for k in SCALARS}
    out["val.mu"] = vals["mu"].astype(np.float32)
    out["val.logvar"] = vals["logvar"].astype(np.float32)
    out["val.fed_tokens"] = vals["fed_tokens"].astype(np.int32)
    out["val.logits_rowsum"] = vals["logits"].sum(-1).astype(np.float64)      # [B,T]
    out["val.logits_absmax"] = np.abs(vals["logits"]).max()
    for k, g in grads.items():
        g64 = g.astype(np.float64)
        out[f"gsum.{k}"] = np.array([g64.sum(), np.abs(g64).sum(), np.abs(g64).max()])
        out[f"grow.{k}"] = g64.reshape(g64.shape[0], -1).sum(1)                # per-row sums
    out["meta"] = np.array([B, T, int(round(tf * 1000))], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, {k: float(vals[k]) for k in SCALARS[:3]})


if __name__ == "__main__":
    torch.set_num_threads(8)
    full_case("tiny_step.npz", TINY, 4, 12, 0.7)
    full_case("small_step.npz", SMALL, 21, 17, 0.5)
    digest_case("default_digest.npz", DEFAULT, 64, 128, 0.9)
