#!/usr/bin/env python3
"""fp64 oracle digests at the FULL sizes of BASELINE.json configs[2] / configs[3] (round 4).

At these sizes the kernel families of the MFMA-bound regime (three-piece tile sweeps, K-split BPTT tile, operand-plane weight
gradients with the bias rider, dense decoder stack) and of the 256-row shard are AUTO-selected by the engine; before this
fixture they met the oracle only at forced small shapes.  One oracle evaluation per case (float64, torch CPU, a few minutes and
up to ~35 GB of host memory each: run them one at a time), stored as digests -- loss scalars, mu / logvar (full up to 512 rows,
else row sums + the first 64 rows), fed-back tokens, per-position logit sums, and for EVERY parameter gradient its sum,
abs-sum, abs-max and per-row sums -- the format of default_digest.npz, compact enough to commit.

    python tests/golden/make_full_size_digests.py [configs2] [default_b256] [default_b2048]     (default: all three)

PARITY UNPINNED against a real MLX run, as every fixture here (the reference cannot run offline: SURVEY.md section 8c); the inputs
are regenerated from the seeds by tests/test_golden_gpu.py, only expected outputs are stored.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import arcvae_oracle as O  # noqa: E402
from helpers import DEFAULT, HYPER, make_case  # noqa: E402

SCALARS = ("total_loss", "recon_loss", "kl_loss", "weighted_kl", "collapse_penalty", "mutual_info", "mi_penalty")
CONFIGS2 = O.Config(vocab_size=80, embedding_dim=128, hidden_dim=512, latent_dim=256, num_conditions=1, num_layers=4)
CASES = {   # name -> (file, config, B, T, teacher-forcing ratio)
    "configs2": ("configs2_digest.npz", CONFIGS2, 512, 128, 0.9),
    "default_b256": ("default_b256_digest.npz", DEFAULT, 256, 128, 0.9),
    "default_b2048": ("default_b2048_digest.npz", DEFAULT, 2048, 128, 0.9),
}


def digest_case(fname, cfg, B, T, tf):
    t0 = time.time()
    params, x, cond, eps, coins = make_case(cfg, B, T, tf)
    vals, grads = O.loss_and_grads(params, cfg, x, cond, eps, coins, dtype=torch.float64, **HYPER)
    out = {f"val.{k}": np.asarray(vals[k], dtype=np.float64) for k in SCALARS}
    for k in ("mu", "logvar"):
        v = np.asarray(vals[k], dtype=np.float64)
        out[f"val.{k}_rowsum"] = v.sum(1)
        out[f"val.{k}_absmax"] = np.abs(v).max()
        out[f"val.{k}"] = v[: (B if B <= 512 else 64)].astype(np.float32)
    out["val.fed_tokens"] = vals["fed_tokens"].astype(np.uint8 if cfg.V <= 255 else np.int32)
    lg = np.asarray(vals["logits"], dtype=np.float64)
    out["val.logits_rowsum"] = lg.sum(-1).astype(np.float32)      # [B,T]
    out["val.logits_absmax"] = np.abs(lg).max()
    for k, g in grads.items():
        g64 = np.asarray(g, dtype=np.float64)
        out[f"gsum.{k}"] = np.array([g64.sum(), np.abs(g64).sum(), np.abs(g64).max()])
        out[f"grow.{k}"] = g64.reshape(g64.shape[0], -1).sum(1)    # per-row sums
    out["meta"] = np.array([B, T, int(round(tf * 1000))], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, {k: float(vals[k]) for k in SCALARS[:3]}, "%.0f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", "8")))
    for name in (sys.argv[1:] or list(CASES)):
        digest_case(*CASES[name])
