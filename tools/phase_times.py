"""Unprofiled per-segment GPU timing of the training step (events around every segment, steady state)."""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mlx-vae_amd", "tests", "oracle"): sys.path.insert(0, os.path.join(ROOT, p))
import torch
import arcvae_hip.engine as E
from helpers import DEFAULT, HYPER, build_engine, make_case
# usage: phase_times.py [big] [bf16] [batch]
from helpers import O
BIG = O.Config(hidden_dim=512, latent_dim=256, num_layers=4)
CFG = BIG if "big" in sys.argv else DEFAULT
BS = next((int(a) for a in sys.argv[1:] if a.isdigit()), 512 if "big" in sys.argv else 64)
params, x, cond, eps, coins = make_case(CFG, BS, 128, 0.9)
eng, enc, dec = build_engine(CFG, params)
if "bf16" in sys.argv:
    eng = E.StepEngine(enc, dec, eng.d, precision="bf16")
ws = eng.workspace(BS, 128)
eng.set_hyper(ws, **HYPER); eng.load_inputs(ws, x, cond, eps, coins)
for _ in range(10): eng.run_step(ws, 2e-4, True)
torch.cuda.synchronize()
orig = E.SegmentRunner.__call__
marks = []
def timed(self, key, fn, stream, **kw):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream); orig(self, key, fn, stream, **kw); e1.record(stream)
    marks.append((key, e0, e1))
E.SegmentRunner.__call__ = timed
acc = collections.OrderedDict(); spans = []
for it in range(20):
    marks.clear()
    s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
    s0.record(); eng.run_step(ws, 2e-4, True); s1.record()
    torch.cuda.synchronize()
    spans.append(s0.elapsed_time(s1) * 1e3)
    for key, e0, e1 in marks:
        acc.setdefault(key, []).append((s0.elapsed_time(e0) * 1e3, s0.elapsed_time(e1) * 1e3))
print("step span us: median %.1f" % sorted(spans)[len(spans)//2])
for key, v in acc.items():
    a = np.median([t[0] for t in v]); b = np.median([t[1] for t in v])
    print(f"{key:12s} start {a:8.1f}  end {b:8.1f}  dur {b-a:7.1f}")
