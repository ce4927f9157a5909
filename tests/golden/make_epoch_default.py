#!/usr/bin/env python3
"""Generate tests/golden/epoch_default.npz: the ELBO components after epochs 0 and 1 of the reference's epoch flow
(trainer.py:177-241) on the SURVEY.md section 8(d) epoch-level workload -- default AR-CVAE (V80 E128 H256 Z128 C1
L2), N = 1000 synthetic rows, 80/10/10 split (train.py:86-96), bs 64, T 128, train.py's argparse hyper-parameters
(lr 2e-4, beta 0 -> 0.05 over 20 epochs, free_bits 1.0, lambda_collapse 0.001, lambda_mi 0.01, 30 epochs).

Produced by the oracle (tests/ref_epoch.py over oracle/arcvae_oracle.py) in float64 -- PARITY UNPINNED against a real
MLX run (the reference cannot run offline and holds no vectors).  Inputs are regenerated from seeds by the test; only
expected outputs are stored.  ~4 minutes on 8 cores.

    python tests/golden/make_epoch_default.py
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
import ref_epoch as R  # noqa: E402

N_ROWS, T, BS, LR, EPOCHS, TOTAL = 1000, 128, 64, 2e-4, 2, 30
HP = dict(beta_start=0.0, beta_end=0.05, warmup=20, lambda_collapse=0.001, free_bits=1.0, lambda_mi=0.01)
KEYS = ("train_loss", "train_recon", "train_kl", "train_collapse", "val_loss", "val_recon", "val_kl", "val_collapse",
        "beta", "teacher_forcing", "mutual_info")


def main():
    cfg = O.Config()
    data = R.synthetic_json(N_ROWS, cfg.V, T)
    tr_i, va_i, _ = R.split_80_10_10(data)
    tr_x, tr_c, mean, std = R.tensorise(data, tr_i)
    va_x, va_c, _, _ = R.tensorise(data, va_i, mean, std)
    p = O.init_params(cfg, 1234)
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(vv) for k, vv in p.items()}
    out = {}
    for epoch in range(EPOCHS):
        np.random.seed(100 + epoch)
        t0 = time.time()
        met = R.reference_epoch(cfg, p, m, v, (tr_x, tr_c), (va_x, va_c), BS, T, LR, epoch, TOTAL, HP, dtype=torch.float64)
        print(f"epoch {epoch}: {time.time() - t0:.0f} s", {k: round(met[k], 6) for k in KEYS}, flush=True)
        out[f"epoch{epoch}"] = np.array([met[k] for k in KEYS], dtype=np.float64)
        out[f"epoch{epoch}.param_l2"] = np.array([float(np.sqrt(np.sum(np.square(w.astype(np.float64))))) for w in p.values()])
    out["keys"] = np.array(KEYS)
    out["param_names"] = np.array(list(p.keys()))
    np.savez_compressed(os.path.join(HERE, "epoch_default.npz"), **out)
    print("wrote epoch_default.npz")


if __name__ == "__main__":
    main()
