"""How long does one hipGraph replay of the training step take on the host vs on the device?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mlx-vae_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from helpers import DEFAULT, HYPER, build_engine, make_case
params, x, cond, eps, coins = make_case(DEFAULT, 64, 128, 0.9)
eng, enc, dec = build_engine(DEFAULT, params)
ws = eng.workspace(64, 128)
eng.set_hyper(ws, **HYPER); eng.load_inputs(ws, x, cond, eps, coins)
for _ in range(3): eng.run_step(ws, 2e-4, True)
torch.cuda.synchronize()
g = list(eng._graphs.values())[0]
for trial in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"replay host {1e3*(t1-t0):.3f} ms, until done {1e3*(t2-t0):.3f} ms")
eng.use_graph = False
for trial in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); eng.run_step(ws, 2e-4, True); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"eager host {1e3*(t1-t0):.3f} ms, until done {1e3*(t2-t0):.3f} ms")
