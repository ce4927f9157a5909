import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mlx-vae_amd", "tests", "oracle"): sys.path.insert(0, os.path.join(ROOT, p))
import torch
from helpers import TINY, HYPER, build_engine, make_case
cfg, B, T = TINY, 8, 12
params, x, cond, eps, coins = make_case(cfg, B, T, 0.7)
for mode in ("eager", "segments"):
    eng, enc, dec = build_engine(cfg, params)
    eng.mode = mode
    ref = None
    for step in range(4):
        out = eng.train_step(x, cond, eps, coins, lr=2e-4, update=False, **HYPER)
        torch.cuda.synchronize()
        g = {n: enc.g(n).cpu().numpy().copy() for n in enc.names()}
        g.update({"dec." + n: dec.g(n).cpu().numpy().copy() for n in dec.names()})
        sc = eng.workspace(B, T).scalars.cpu().numpy()[:3]
        if ref is None: ref = g; print(mode, "step0 loss", sc)
        else:
            bad = {n: float(np.abs(g[n] - ref[n]).max() / (np.abs(ref[n]).max() + 1e-30)) for n in g}
            worst = sorted(bad.items(), key=lambda kv: -kv[1])[:4]
            print(mode, "step", step, "loss", sc, "worst grad diffs", worst)
