"""The fused seam kernel alone on the device (no decoder / weight-gradient streams beside it): us per launch for phases 3 / 1 / 2,
against the five launches it replaces.  usage: python tools/r4_seam_alone.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mlx-vae_amd", "tests", "oracle"): sys.path.insert(0, os.path.join(ROOT, p))
import torch
import arcvae_hip.engine as E
from helpers import DEFAULT, HYPER, build_engine, make_case
B, T = 64, 128
params, x, cond, eps, coins = make_case(DEFAULT, B, T, 0.9)
eng, enc, dec = build_engine(DEFAULT, params)
ws = eng.workspace(B, T)
eng.set_hyper(ws, **HYPER); eng.load_inputs(ws, x, cond, eps, coins)
for _ in range(3): eng.run_step(ws, 2e-4, False)
torch.cuda.synchronize()
fb = float(eng.hyper_host["free_bits"])
def rearm():
    ws.psync[4864:5184].zero_(); ws.stats.zero_()
def timed(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
base = timed(rearm)
def seam(ph):
    def f():
        rearm(); E.encoder_seam(enc, ws, eng.d, fb, ph); ws.seam_bwd_done = False
    return f
def seam12():
    rearm(); E.encoder_seam(enc, ws, eng.d, fb, 1); E.encoder_seam(enc, ws, eng.d, fb, 2); ws.seam_bwd_done = False
def launches():
    rearm()
    E.call("arcvae_enc_heads_forward", E.ptr(ws.hseq[1, T - 1]), E.ptr(ws.cond), E.ptr(enc.p("condition_fc.weight")),
           E.ptr(enc.p("condition_fc.bias")), E.ptr(enc.p("fc_mu.weight")), E.ptr(enc.p("fc_mu.bias")),
           E.ptr(enc.p("fc_logvar_hidden.weight")), E.ptr(enc.p("fc_logvar_hidden.bias")), E.ptr(enc.p("fc_logvar.weight")),
           E.ptr(enc.p("fc_logvar.bias")), E.ptr(ws.eps), E.ptr(ws.comb), E.ptr(ws.lh), E.ptr(ws.mu_raw), E.ptr(ws.lv_raw), E.ptr(ws.mu),
           E.ptr(ws.logvar), E.ptr(ws.z), E.ptr(ws.stats), B, 256, 128, 1, fb, 1, E.stream_ptr())
    E.latent_loss(ws, eng.d, fb, True)
    E.EncoderBackwardPlan(enc, ws, eng.d).heads(1)
def graphed(fn, n=100):
    """device time per iteration: the launches captured once, replayed n times back to back"""
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
gb = graphed(rearm)
print("captured: re-arm fills %.1f us; five launches %.1f us" % (gb, graphed(launches) - gb))
for d in range(0, 10):
    os.environ["ARCVAE_SEAM_DEBUG"] = str(d)
    print("captured: seam phases=3, stop point %d: %.1f us" % (d, graphed(seam(3)) - gb))
os.environ["ARCVAE_SEAM_DEBUG"] = "0"
print("captured: seam phases=1 %.1f us, phases=1 then 2 %.1f us" % (graphed(seam(1)) - gb, graphed(seam12) - gb))
print("re-arm fills alone: %.1f us" % base)
for name, fn in (("seam phases=3", seam(3)), ("seam phases=1", seam(1)), ("seam phases=1 then 2", seam12), ("five launches", launches)):
    print("%-24s %.1f us (minus fills: %.1f)" % (name, timed(fn), timed(fn) - base))
