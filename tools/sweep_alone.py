"""The two encoder sweeps ALONE on the device (no decoder / weight-gradient GEMMs beside them): us per tick, forward and BPTT.
usage: sweep_alone.py [batch] ; environment knobs select the kernel family (ARCVAE_PERSIST_GROUPS, ARCVAE_PERSIST2_ASSIGN, ...)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mlx-vae_amd", "tests", "oracle"): sys.path.insert(0, os.path.join(ROOT, p))
import torch
import arcvae_hip.engine as E
from arcvae_hip import _lib
from helpers import DEFAULT, HYPER, build_engine, make_case
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = 128
params, x, cond, eps, coins = make_case(DEFAULT, BS, T, 0.9)
eng, enc, dec = build_engine(DEFAULT, params)
ws = eng.workspace(BS, T)
eng.set_hyper(ws, **HYPER); eng.load_inputs(ws, x, cond, eps, coins)
eng.run_step(ws, 2e-4, False); torch.cuda.synchronize(); eng.check_gates()
d = eng.d
plan = E.EncoderBackwardPlan(enc, ws, d)
S = T + 2 * (d.L - 1)

def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))

fwd = timed(lambda: E.encoder_forward(enc, ws, d, 1.0))
bwd = timed(lambda: plan.sweep(0, S, None, 0))
err = int(ws.psync[500].item())
print(f"bs {BS} groups {_lib.load().arcvae_enc_lstm_persist_groups(BS, d.H, d.L)} persistent fwd {E.persistent_forward_ok(ws, d)} "
      f"bwd-rs {E.bptt_reduce_scatter_ok(ws, d)}: forward (prologue + table0 + sweep + heads) {fwd:.1f} us = {fwd / (T + d.L - 1):.2f} us/tick; "
      f"BPTT sweep {bwd:.1f} us = {bwd / S:.2f} us/tick; err word {err}")
