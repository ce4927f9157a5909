"""Would two independent half-batch chains on two hardware queues finish sooner than one full-batch chain?
(Rows of the LSTM recurrence are independent.)  Forward and BPTT sweeps, B = 64 on one stream vs 2 x B = 32 on two."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mlx-vae_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as Bn
from arcvae_hip import _lib, engine as E
from arcvae_hip.store import ParamStore, decoder_shapes, encoder_shapes
dev = torch.device("cuda", 0); torch.cuda.set_device(0); _lib.load()
V, EMB, H, Z, C, L, T = Bn.V, Bn.EMB, Bn.H, Bn.Z, Bn.C, Bn.L, Bn.T
gen = torch.Generator().manual_seed(1234)
enc = ParamStore(encoder_shapes(V, EMB, H, Z, C, L), dev); dec = ParamStore(decoder_shapes(V, EMB, H, Z, C, L), dev)
enc.init_mlx_like(H, gen); dec.init_mlx_like(H, gen)
eng = E.StepEngine(enc, dec, E.ModelDims(V=V, E=EMB, H=H, Z=Z, C=C, L=L))
wx, _k1 = E._layer_ptrs(enc, L, "Wx", skip0=True); wh, _k2 = E._layer_ptrs(enc, L, "Wh"); bs, _k3 = E._layer_ptrs(enc, L, "bias", skip0=True)

def make(B):
    ws = E.Workspace(eng.d, B, T, dev, True)
    rs = np.random.RandomState(B); x, cond = Bn.synth(rs, B)
    ws.x.copy_(torch.tensor(x)); ws.x_tb.copy_(torch.tensor(x).t().contiguous()); ws.table0.normal_(); ws.dcomb.normal_()
    return ws
def fwd(ws): E.call("arcvae_enc_lstm_forward", E.ptr(ws.x_tb), E.ptr(ws.table0), wx, wh, bs, E.ptr(ws.hseq), E.ptr(ws.hseq_t), E.ptr(ws.cseq), E.ptr(ws.gseq), E.ptr(ws.wt), E.ptr(ws.wT), ws.B, T, V, H, L, 0, None, None, E.stream_ptr())
def bwd(ws): E.call("arcvae_enc_lstm_backward", wx, wh, E.ptr(ws.cseq), E.ptr(ws.gseq), E.ptr(ws.dcomb), 2 * H, E.ptr(ws.dG), E.ptr(ws.dG_t), E.ptr(ws.dcs), E.ptr(ws.dxs), E.ptr(ws.wT), ws.B, T, H, L, 0, T + 2 * (L - 1), 0, None, None, None, E.stream_ptr())

def graph(fn, ws, stream):
    with torch.cuda.stream(stream):
        fn(ws)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        fn(ws)
    return g

def timed(graphs_streams, reps=20):
    main = torch.cuda.current_stream()
    def once():
        ev = torch.cuda.Event(); ev.record(main)
        for g, st in graphs_streams:
            st.wait_event(ev)
            with torch.cuda.stream(st):
                g.replay()
        for g, st in graphs_streams:
            main.wait_stream(st)
    once(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(reps): once()
    e1.record(main); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps

s = [torch.cuda.Stream() for _ in range(4)]
for name, fn in (("forward", fwd), ("BPTT", bwd)):
    w64 = make(64); fwd(w64); torch.cuda.synchronize()
    one = timed([(graph(fn, w64, s[0]), s[0])])
    halves = [make(32), make(32)]
    for w in halves: fwd(w)
    torch.cuda.synchronize()
    two = timed([(graph(fn, halves[0], s[0]), s[0]), (graph(fn, halves[1], s[1]), s[1])])
    one32 = timed([(graph(fn, halves[0], s[0]), s[0])])
    quarters = [make(16) for _ in range(4)]
    for w in quarters: fwd(w)
    torch.cuda.synchronize()
    four = timed([(graph(fn, quarters[i], s[i]), s[i]) for i in range(4)])
    print(f"{name}: one chain B=64 {one:.1f} us | one chain B=32 {one32:.1f} us | two chains 2xB=32 on two streams {two:.1f} us | four chains 4xB=16 {four:.1f} us")
