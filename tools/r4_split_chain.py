"""Round 4 probe: the launch-based BPTT sweep of the 256-row shard as ONE chain of 130 launches over 256 rows, against TWO / FOUR
independent chains over 128 / 64 rows on separate streams (rows of the recurrence are independent; each chain's launches are half
/ quarter the size and the chains' launch seams and cold fetches could overlap).  ARCVAE_PERSIST=0 ARCVAE_STEP_TILE=22 force the
same mid-batch step kernel (lstm_bwd_step2_kernel) for every size.  usage: python tools/r4_split_chain.py [rows=256]"""
import os, sys
os.environ["ARCVAE_PERSIST"] = "0"; os.environ["ARCVAE_STEP_TILE"] = "22"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mlx-vae_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as Bn
from arcvae_hip import _lib, engine as E
from arcvae_hip.store import ParamStore, decoder_shapes, encoder_shapes
dev = torch.device("cuda", 0); torch.cuda.set_device(0); _lib.load()
V, EMB, H, Z, C, L, T = Bn.V, Bn.EMB, Bn.H, Bn.Z, Bn.C, Bn.L, Bn.T
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
gen = torch.Generator().manual_seed(1234)
enc = ParamStore(encoder_shapes(V, EMB, H, Z, C, L), dev); dec = ParamStore(decoder_shapes(V, EMB, H, Z, C, L), dev)
enc.init_mlx_like(H, gen); dec.init_mlx_like(H, gen)
eng = E.StepEngine(enc, dec, E.ModelDims(V=V, E=EMB, H=H, Z=Z, C=C, L=L))
wx, _k1 = E._layer_ptrs(enc, L, "Wx", skip0=True); wh, _k2 = E._layer_ptrs(enc, L, "Wh"); bs, _k3 = E._layer_ptrs(enc, L, "bias", skip0=True)
S = T + 2 * (L - 1)

def make(B):
    ws = eng.workspace(B, T, True) if False else E.Workspace(eng.d, B, T, dev, True)
    rs = np.random.RandomState(B); x, cond = Bn.synth(rs, B)
    ws.x.copy_(torch.tensor(x)); ws.x_tb.copy_(torch.tensor(x).t().contiguous()); ws.table0.normal_(); ws.dcomb.normal_()
    return ws
def fwd(ws): E.call("arcvae_enc_lstm_forward", E.ptr(ws.x_tb), E.ptr(ws.table0), wx, wh, bs, E.ptr(ws.hseq), E.ptr(ws.hseq_t), E.ptr(ws.cseq), E.ptr(ws.gseq), E.ptr(ws.wt), E.ptr(ws.wT), ws.B, T, V, H, L, 0, E._caps(ws), None, None, E.stream_ptr())
def bwd(ws): E.call("arcvae_enc_lstm_backward", wx, wh, E.ptr(ws.cseq), E.ptr(ws.gseq), E.ptr(ws.dcomb), 2 * H, E.ptr(ws.dG), E.ptr(ws.dG_t), E.ptr(ws.dcs), E.ptr(ws.dxs), E.ptr(ws.wT), ws.B, T, H, L, 0, S, 0, E._caps(ws), None, None, None, E.stream_ptr())

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
for name, fn in (("BPTT", bwd), ("forward", fwd)):
    res = []
    for parts in (1, 2, 4):
        wss = [make(R // parts) for _ in range(parts)]
        for w in wss: fwd(w)
        torch.cuda.synchronize()
        t = timed([(graph(fn, wss[i], s[i]), s[i]) for i in range(parts)])
        res.append(f"{parts} chain(s) x {R // parts} rows: {t:.0f} us = {t / S:.2f} us per tick")
        if parts > 1:
            t1 = timed([(graph(fn, wss[0], s[0]), s[0])])
            res.append(f"(one {R // parts}-row chain alone: {t1:.0f} us)")
        del wss
    print(f"{name} sweep, launch-based step kernels, {R} rows: " + " | ".join(res))
