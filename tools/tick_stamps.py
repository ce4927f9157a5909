"""Where a tick of the persistent sweeps goes: the diagnostic library (tools/build_stamps.sh) writes eight stamps per tick of
one block; this prints the median gap between consecutive stamps (us) over the steady-state ticks, the sweeps run ALONE.
usage: ARCVAE_HIP_LIB=ab_libs/libarcvae_stamps.so python tools/tick_stamps.py [batch]
forward stamps: 0 loop top | 1 flags of the previous tick seen (poll + barrier) | 2 operand loads issued | 3 products done
                (LDS partials written) | 4 block barrier | 5 cell epilogue done, stores issued | 6 stores acknowledged | 7 barrier + flag
BPTT stamps:    0 loop top | 1 epilogue operands requested | 2 products done, partial stores issued | 3 stores acknowledged |
                4 flag + poll + barrier (all partials of the tick published) | 5 first group's gather returned | 6 all epilogues | 7 barrier"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("ARCVAE_HIP_LIB", os.path.join(ROOT, "ab_libs", "libarcvae_stamps.so"))   # the diagnostic build
for p in ("mlx-vae_amd", "tests", "oracle"): sys.path.insert(0, os.path.join(ROOT, p))
import torch
import arcvae_hip.engine as E
from arcvae_hip import _lib
from helpers import DEFAULT, HYPER, build_engine, make_case
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = 128
params, x, cond, eps, coins = make_case(DEFAULT, BS, T, 0.9)
eng, enc, dec = build_engine(DEFAULT, params)
ws = eng.workspace(BS, T)
ws.trace_fwd = torch.zeros(8 * (T + 8), dtype=torch.int64, device=eng.device)
ws.trace_bwd = torch.zeros(8 * (T + 8), dtype=torch.int64, device=eng.device)
eng.set_hyper(ws, **HYPER); eng.load_inputs(ws, x, cond, eps, coins)
eng.mode = "eager"
eng.run_step(ws, 2e-4, False); torch.cuda.synchronize(); eng.check_gates()
d = eng.d
plan = E.EncoderBackwardPlan(enc, ws, d)
S = T + 2 * (d.L - 1)

def report(name, buf, n):
    st = buf.cpu().numpy().reshape(-1, 8)[:n].astype(np.float64) / 100.0   # us
    lo, hi = 8, n - 8
    gaps = np.diff(st[lo:hi], axis=1)                      # 7 gaps inside a tick
    wrap = st[lo + 1:hi + 1, 0] - st[lo:hi, 7]              # end of tick -> top of the next
    tick = st[lo + 1:hi + 1, 1] - st[lo:hi, 1]
    print(f"{name}: tick {np.median(tick):.2f} us | gaps 0-1 .. 6-7:", " ".join(f"{g:.2f}" for g in np.median(gaps, axis=0)),
          f"| 7-0' {np.median(wrap):.2f}")

for rep in range(3):
    E.encoder_forward(enc, ws, d, 1.0); torch.cuda.synchronize()
    report("forward", ws.trace_fwd, T + d.L - 1)
for rep in range(3):
    plan.sweep(0, S, None, 0); torch.cuda.synchronize()
    report("BPTT   ", ws.trace_bwd, S)
print("err word", int(ws.psync[500].item()), "groups", _lib.load().arcvae_enc_lstm_persist_groups(BS, d.H, d.L))
