"""Repeat one training step (no update) many times and look for steps whose encoder gradients / gate gradients differ
from the majority: frequency and location of an intermittent ordering error.  usage: [RH_B=64 RH_T=12 RH_L=2] python tools/race_hunt.py [N]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mlx-vae_amd", "tests", "oracle"): sys.path.insert(0, os.path.join(ROOT, p))
import torch
import arcvae_oracle as O
from helpers import HYPER, build_engine, make_case
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
H, L, B, T, C = 256, int(os.environ.get("RH_L", 2)), int(os.environ.get("RH_B", 64)), int(os.environ.get("RH_T", 12)), 1
cfg = O.Config(vocab_size=60, embedding_dim=32, hidden_dim=H, latent_dim=16, num_conditions=C, num_layers=L)
params, x, cond, eps, coins = make_case(cfg, B, T, 0.6)
eng, enc, dec = build_engine(cfg, params)
names = ["lstm_layer_0.Wh", "lstm_layer_0.bias", "embedding.weight"] + [f"lstm_layer_{l}.{w}" for l in range(1, L) for w in ("Wh", "Wx")]
ref, refdg, bad = None, None, []
for it in range(N):
    eng.train_step(x, cond, eps, coins, lr=2e-4, update=False, **HYPER)
    torch.cuda.synchronize()
    ws = eng.workspace(B, T, True)
    g = {n: enc.g(n).clone() for n in names}
    dg = ws.dG.clone()
    hs = ws.hseq.clone()
    if ref is None:
        ref, refdg, refh, nh = g, dg, hs, 0
        continue
    if not torch.equal(hs, refh):     # the forward sweep is deterministic: any difference is an ordering error
        nh += 1
        if nh <= 3:
            dh = (hs - refh).abs()
            print(f"step {it}: hseq differs, max {float(dh.max()):.3e}, n={int((dh > 0).sum())}")
    errs = {n: float((g[n] - ref[n]).abs().max() / ref[n].abs().max()) for n in names}
    if max(errs.values()) > 1e-5:
        d = (dg - refdg).abs().reshape(L, T, B, 4 * H)
        idx = np.unravel_index(int(d.argmax()), d.shape)
        nz = (d > 1e-7 * float(refdg.abs().max())).nonzero()
        ts = sorted(set((int(a), int(b)) for a, b in zip(nz[:, 0].tolist(), nz[:, 1].tolist())))
        rows = sorted(set(nz[:, 2].tolist()))
        bad.append(it)
        if len(bad) <= 4:
            print(f"step {it}: " + " ".join(f"{n}={e:.1e}" for n, e in errs.items()))
            print(f"   dG max diff {float(d.max()):.3e} (ref max {float(refdg.abs().max()):.3e}) at (l,t,b,col)={idx}; "
                  f"differing (l,t): {ts[:12]}  rows: {rows[:16]} n={int(nz.shape[0])}")
eng.check_gates()
print(f"{len(bad)} deviating steps of {N - 1}: {bad[:20]};  steps with a different hseq: {nh}")
