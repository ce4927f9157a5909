"""Two-group persistent sweeps at bs 256: run steps one by one, stop at the first raised error word, dump the sync words."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mlx-vae_amd", "tests", "oracle"): sys.path.insert(0, os.path.join(ROOT, p))
import torch
import arcvae_hip.engine as E
from helpers import DEFAULT, HYPER, build_engine, make_case
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
params, x, cond, eps, coins = make_case(DEFAULT, BS, 128, 0.9)
eng, enc, dec = build_engine(DEFAULT, params)
ws = eng.workspace(BS, 128)
eng.set_hyper(ws, **HYPER); eng.load_inputs(ws, x, cond, eps, coins)
import time
for it in range(N):
    t0 = time.time()
    eng.run_step(ws, 2e-4, True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    ps = ws.psync.cpu().numpy().astype(np.int64)
    if ps[500] != 0 or dt > 0.5 or it == N - 1:
        print(f"step {it}: err={ps[500]} wall={dt*1e3:.1f} ms gates_err={eng.gates.errors() if eng.gates else None}")
        print(" fwd role counters [g][xcc]:", ps[1536:1552].reshape(2, 8).tolist())
        print(" bwd role counters chunk0:", ps[2064:2080].reshape(2, 8).tolist(), "chunk1:", ps[2080:2096].reshape(2, 8).tolist())
        cu = ps[2304:4352].reshape(8, 256)
        vals, cnts = np.unique(cu, return_counts=True)
        print(" CU table histogram:", dict(zip(vals.tolist(), cnts.tolist())))
        ff = ps[1024:1536].reshape(2, 8, 32); bf = ps[1552:2064].reshape(2, 8, 32)
        print(" fwd flags min/max per group:", [(int(ff[g].min()), int(ff[g].max())) for g in range(2)])
        print(" bwd flags min/max per group:", [(int(bf[g].min()), int(bf[g].max())) for g in range(2)])
        if ps[500] != 0:
            break
