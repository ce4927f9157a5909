#!/usr/bin/env python3
"""BASELINE.json configs[4]: greedy sampling of 10 000 molecules (models/decoder_sampling.py) on 1 MI355X.

10 batches of bs 1024 (last one 784 rows), max_length 80 (the API default) and 128, early stopping on/off.
One batch = one dense decoder pass over B*V rows + one table walk, replayed as a captured hipGraph.
Prints molecules/s and tokens/s; untrained random weights (tokens after EOS are generated, Q9).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mlx-vae_amd"))
import torch  # noqa: E402
from models.vae import ARCVAE  # noqa: E402

vae = ARCVAE(vocab_size=80, embedding_dim=128, hidden_dim=256, latent_dim=128, num_conditions=1, num_layers=2)
rs = np.random.RandomState(0)
conds = [torch.tensor(rs.standard_normal((b, 1)).astype(np.float32)).cuda() for b in [1024] * 9 + [784]]
out = {}
for max_len in (80, 128):
    for early in (True, False):
        for c in conds[:1] + conds[-1:]:  # warm-up / capture both batch shapes
            vae.decoder_sampling.generate_with_temperature(None if False else torch.zeros(c.shape[0], 128), c,
                                                           max_length=max_len, early_stopping=early)
        zs = [torch.zeros(c.shape[0], 128, device="cuda") for c in conds]  # z is accepted and unused (Q2)
        times = []
        for rep in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ntok = 0
            for c, z in zip(conds, zs):
                toks = vae.decoder_sampling.generate_with_temperature(z, c, max_length=max_len, early_stopping=early)
                ntok += toks.numel()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        dt = sorted(times)[len(times) // 2]
        out[f"max_length={max_len},early_stopping={early}"] = {
            "molecules_per_s": 10000 / dt, "tokens_per_s": ntok / dt, "ms_per_10k_median": 1e3 * dt,
            "ms_per_10k_all": [round(1e3 * t, 2) for t in times]}
print(json.dumps({"metric": "greedy sampling, 10k molecules, bs 1024, 1x MI355X, fp32", "results": out}))
