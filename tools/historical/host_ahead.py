"""Is the host ahead of the GPU in steady state?  For each step: was the previous step's end event already
complete when the host STARTED enqueuing this step, and how long did each enqueue phase take on the host."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mlx-vae_amd", "tests", "oracle"): sys.path.insert(0, os.path.join(ROOT, p))
import torch
from helpers import DEFAULT, HYPER, build_engine, make_case
params, x, cond, eps, coins = make_case(DEFAULT, 64, 128, 0.9)
eng, enc, dec = build_engine(DEFAULT, params)
ws = eng.workspace(64, 128)
eng.set_hyper(ws, **HYPER); eng.load_inputs(ws, x, cond, eps, coins)
for _ in range(5): eng.run_step(ws, 2e-4, True)
torch.cuda.synchronize()
evs = []
done_at_start = []
t_host = []
for i in range(30):
    t0 = time.perf_counter()
    if evs: done_at_start.append(evs[-1].query())
    eng.run_step(ws, 2e-4, True)
    e = torch.cuda.Event(); e.record(); evs.append(e)
    t_host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("prev step already finished when host started the next:", sum(done_at_start), "of", len(done_at_start))
print("host enqueue ms per step: median %.3f" % (1e3 * sorted(t_host)[len(t_host)//2]))
# fine-grained: time each segment launch on the host for one step
import arcvae_hip.engine as E
orig = E.SegmentRunner.__call__
log = []
def timed(self, key, fn, stream):
    t0 = time.perf_counter(); orig(self, key, fn, stream); log.append((key, 1e6 * (time.perf_counter() - t0)))
E.SegmentRunner.__call__ = timed
for _ in range(3):
    log.clear(); t0 = time.perf_counter(); eng.run_step(ws, 2e-4, True); tot = time.perf_counter() - t0
print("segments (us on host):", [(k, round(v)) for k, v in log], "total", round(1e6 * tot))
torch.cuda.synchronize()
