"""Worst element-wise deviation (tests/helpers.py: elem_err) of the HIP step and of the fp32 oracle against the fp64
oracle, per tensor class, over the parity cases -- the measurement behind ELEM_ATOL_* in tests/helpers.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mlx-vae_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import arcvae_oracle as O
from helpers import DEFAULT, HYPER, SMALL, TINY, build_engine, make_case, elem_err, rel_err

CASES = [("tiny", TINY, 4, 12, 0.7), ("small", SMALL, 21, 17, 0.5), ("tiny_tf0", TINY, 5, 16, 0.0),
         ("h512l4", O.Config(80, 32, 512, 16, 1, 4), 24, 10, 0.7), ("default", DEFAULT, 64, 128, 0.9)]
for name, cfg, B, T, tf in CASES:
    params, x, cond, eps, coins = make_case(cfg, B, T, tf)
    v64, g64 = O.loss_and_grads(params, cfg, x, cond, eps, coins, dtype=torch.float64, **HYPER)
    v32, g32 = O.loss_and_grads(params, cfg, x, cond, eps, coins, dtype=torch.float32, **HYPER)
    eng, enc, dec = build_engine(cfg, params)
    out = eng.train_step(x, cond, eps, coins, lr=2e-4, update=False, **HYPER)
    torch.cuda.synchronize()
    ws = eng.workspace(B, T)
    hip = {"mu": out["mu"].cpu().numpy(), "logvar": out["logvar"].cpu().numpy(), "z": out["z"].cpu().numpy(),
           "logits": eng.gather_logits(ws).cpu().numpy()}
    print(f"== {name}: B {B} T {T}")
    for k in ("mu", "logvar", "z", "logits"):
        for af in (1e-6, 1e-5):
            print(f"  fwd {k:8s} atol {af:g}: hip {elem_err(hip[k], v64[k], 1e-4, af)[0]:8.3f}  oracle32 {elem_err(v32[k], v64[k], 1e-4, af)[0]:8.3f}"
                  f"   normwise hip {rel_err(hip[k], v64[k]):.2e}")
    rows = []
    for n, g in g64.items():
        if np.abs(g).max() == 0: continue
        mod, pn = n.split(".", 1)
        got = (enc if mod == "encoder" else dec).g(pn).cpu().numpy()
        rows.append((n, elem_err(got, g, 1e-4, 1e-6)[0], elem_err(g32[n], g, 1e-4, 1e-6)[0],
                     elem_err(got, g, 1e-4, 1e-5)[0], elem_err(g32[n], g, 1e-4, 1e-5)[0], rel_err(got, g)))
    for r in rows:
        print(f"  grad {r[0]:36s} atol 1e-6: hip {r[1]:8.3f} oracle32 {r[2]:8.3f} | atol 1e-5: hip {r[3]:7.3f} oracle32 {r[4]:7.3f} | normwise {r[5]:.2e}")
    print(f"  WORST grads: atol 1e-6 hip {max(r[1] for r in rows):.3f} oracle32 {max(r[2] for r in rows):.3f}; atol 1e-5 hip {max(r[3] for r in rows):.3f} oracle32 {max(r[4] for r in rows):.3f}")
