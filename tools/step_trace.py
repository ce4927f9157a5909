"""Device-side launch trace of the LSTM sweeps inside a real training step (no profiler attached):
python tools/step_trace.py [--batch 64] [--steps 30]
Every step launch stamps the 100 MHz wall clock at the start and end of its block (0,0,0)
(the `trace` argument of the sweep entry points).  Prints cadence (start-to-start) statistics, the gaps, and a coarse timeline."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mlx-vae_amd"))
sys.path.insert(0, ROOT)
import torch
import bench as Bn
from arcvae_hip import _lib, engine as E
from arcvae_hip.store import ParamStore, decoder_shapes, encoder_shapes

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--H", type=int, default=Bn.H)
ap.add_argument("--L", type=int, default=Bn.L)
ap.add_argument("--Z", type=int, default=Bn.Z)
ap.add_argument("--dp", action="store_true", help="trace the data-parallel driver (RCCL group of 1 rank, forced collectives)")
ap.add_argument("--only-step", action="store_true", help="skip the calibration / hypothesis sections")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
_lib.load()
V, EMB, C, T = Bn.V, Bn.EMB, Bn.C, Bn.T
H, L, Z = args.H, args.L, args.Z
B = args.batch
dims = E.ModelDims(V=V, E=EMB, H=H, Z=Z, C=C, L=L)
gen = torch.Generator().manual_seed(1234)
enc = ParamStore(encoder_shapes(V, EMB, H, Z, C, L), dev)
dec = ParamStore(decoder_shapes(V, EMB, H, Z, C, L), dev)
enc.init_mlx_like(H, gen); dec.init_mlx_like(H, gen)
eng = E.StepEngine(enc, dec, dims)
ws = eng.workspace(B, T, train=True)
eng.set_hyper(ws, **Bn.HYPER)
rs = np.random.RandomState(67)
x, cond = Bn.synth(rs, B)
ws.x.copy_(torch.tensor(x)); ws.cond.copy_(torch.tensor(cond))
ws.eps.copy_(torch.tensor(rs.standard_normal((B, Z)).astype(np.float32)))
cap = 2 * (T + 2 * L + 4)
buf = torch.zeros(2 * cap, dtype=torch.int64, device=dev)
ws.trace_fwd, ws.trace_bwd = buf[:cap], buf[cap:]   # forward launch s -> slot s, BPTT launch s -> slot cap/2 + s
if args.dp:
    import torch.distributed as dist
    from arcvae_hip.dp import DataParallelStep, EngineOps
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    os.environ["ARCVAE_DP_FORCE_COLLECTIVES"] = "1"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dp = DataParallelStep(EngineOps(eng, ws, Bn.LR, B, use_graph=True))
    print("dp gated:", dp.ops.gated)
    step_fn = dp.step
else:
    step_fn = lambda: eng.run_step(ws, Bn.LR, update=True)
for _ in range(args.steps):
    step_fn()
torch.cuda.synchronize()
eng.check_gates()
t = buf.cpu().numpy().reshape(cap, 2).astype(np.float64) / 100.0  # us
nf, nb = T + L - 1, T + 2 * (L - 1)
fw, bw = t[:nf], t[cap // 2: cap // 2 + nb]
t0 = fw[0, 0]
fw, bw = fw - t0, bw - t0
def stats(name, a):
    per = np.diff(a[:, 0])
    dur = a[:, 1] - a[:, 0]
    print(f"{name}: {len(a)} launches, span {a[-1,1]-a[0,0]:.1f} us; start-to-start median {np.median(per):.2f} mean {per.mean():.2f} "
          f"p90 {np.percentile(per,90):.2f} max {per.max():.2f}; block(0,0,0) life median {np.median(dur):.2f}")
    big = np.where(per > 2.0 * np.median(per))[0]
    print(f"  periods > 2x median: {len(big)}, extra time {np.sum(per[big]-np.median(per)):.1f} us; at launches {big[:20].tolist()} -> {np.round(per[big][:20],1).tolist()}")
    for lo in range(0, len(a) - 1, 16):
        seg = per[lo:lo + 16]
        print(f"  launches {lo:3d}-{lo+len(seg)-1:3d}: t={a[lo,0]:7.1f} us  mean period {seg.mean():5.2f}  max {seg.max():5.2f}")
stats("forward sweep", fw)
stats("BPTT sweep", bw)
print(f"forward end -> BPTT start: {bw[0,0]-fw[-1,1]:.1f} us;  step (first fwd launch -> last bwd launch end): {bw[-1,1]:.1f} us")
if args.dp:
    dist.destroy_process_group()
if args.only_step or args.dp:
    sys.exit(0)

# ---- calibration: the same sweeps replayed ALONE (one linear graph each, nothing else on the chip) ----
def alone(name, fn, n, off):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=st):
        fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    a = buf.cpu().numpy().reshape(cap, 2).astype(np.float64)[off:off + n] / 100.0
    a = a - a[0, 0]
    stats(name + " ALONE (graph replay)", a)

wx, _k1 = E._layer_ptrs(enc, L, "Wx", skip0=True)
wh, _k2 = E._layer_ptrs(enc, L, "Wh")
bs, _k3 = E._layer_ptrs(enc, L, "bias", skip0=True)
alone("forward sweep", lambda: E.call("arcvae_enc_lstm_forward", E.ptr(ws.x_tb), E.ptr(ws.table0), wx, wh, bs, E.ptr(ws.hseq),
      E.ptr(ws.hseq_t), E.ptr(ws.cseq), E.ptr(ws.gseq), E.ptr(ws.wt), E.ptr(ws.wT), B, T, V, H, L, 0, E._caps(ws), None, E.ptr(ws.trace_fwd), E.stream_ptr()), nf, 0)
alone("BPTT sweep", lambda: E.call("arcvae_enc_lstm_backward", wx, wh, E.ptr(ws.cseq), E.ptr(ws.gseq), E.ptr(ws.dcomb), 2 * H,
      E.ptr(ws.dG), E.ptr(ws.dG_t), E.ptr(ws.dcs), E.ptr(ws.dxs), E.ptr(ws.wT), B, T, H, L, 0, nb, 0, E._caps(ws), None, None, E.ptr(ws.trace_bwd), E.stream_ptr()), nb, cap // 2)

# ---- hypothesis checks -------------------------------------------------------------------------------
big = torch.empty(160 * 1024 * 1024, dtype=torch.float32, device=dev)  # 640 MB: evicts the 256 MB Infinity Cache
def alone_cold(name, fn, n, off):
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=st):
        big.fill_(1.0)
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a = buf.cpu().numpy().reshape(cap, 2).astype(np.float64)[off:off + n] / 100.0
    stats(name + " ALONE after a 640 MB write (Infinity Cache evicted)", a - a[0, 0])
alone_cold("forward sweep", lambda: E.call("arcvae_enc_lstm_forward", E.ptr(ws.x_tb), E.ptr(ws.table0), wx, wh, bs, E.ptr(ws.hseq),
      E.ptr(ws.hseq_t), E.ptr(ws.cseq), E.ptr(ws.gseq), E.ptr(ws.wt), E.ptr(ws.wT), B, T, V, H, L, 0, E._caps(ws), None, E.ptr(ws.trace_fwd), E.stream_ptr()), nf, 0)
alone_cold("BPTT sweep", lambda: E.call("arcvae_enc_lstm_backward", wx, wh, E.ptr(ws.cseq), E.ptr(ws.gseq), E.ptr(ws.dcomb), 2 * H,
      E.ptr(ws.dG), E.ptr(ws.dG_t), E.ptr(ws.dcs), E.ptr(ws.dxs), E.ptr(ws.wT), B, T, H, L, 0, nb, 0, E._caps(ws), None, None, E.ptr(ws.trace_bwd), E.stream_ptr()), nb, cap // 2)

# the engine's own forward segment, replayed with nothing else in flight
run = E.SegmentRunner(True)
for _ in range(6):
    eng.enqueue_encoder_forward(ws, run)
torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(cap, 2).astype(np.float64)[:nf] / 100.0
stats("engine enc_fwd segment, nothing else in flight", a - a[0, 0])

# ---- does a BLOCKED second/third queue slow the dependent chain on the main queue? ----
def blocked_queue_test(nblocked):
    others = [torch.cuda.Stream() for _ in range(nblocked)]
    main = torch.cuda.current_stream()
    dummy = torch.zeros(1024, device=dev)
    for _ in range(6):
        ev0 = torch.cuda.Event(); ev0.record(main)
        eng.enqueue_encoder_forward(ws, run)          # replayed segment on main
        ev = torch.cuda.Event(); ev.record(main)
        for o in others:                              # each other queue: a wait that stays blocked for the whole sweep
            o.wait_event(ev)
            with torch.cuda.stream(o):
                dummy.add_(1.0)
        for o in others:
            main.wait_stream(o)
    torch.cuda.synchronize()
    a = buf.cpu().numpy().reshape(cap, 2).astype(np.float64)[:nf] / 100.0
    stats(f"engine enc_fwd segment with {nblocked} other queue(s) blocked on its end event", a - a[0, 0])
for nb_ in (1, 2, 3):
    blocked_queue_test(nb_)

# ---- the same, but the other queues wait in a polling GATE KERNEL instead of on an event ----
def gated_queue_test(nblocked):
    others = [torch.cuda.Stream() for _ in range(nblocked)]
    main = torch.cuda.current_stream()
    flag = torch.zeros(64, dtype=torch.int32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t_wall = time.perf_counter()
    for it in range(6):
        for o in others:
            with torch.cuda.stream(o):
                _lib.call("arcvae_gate_wait", _lib.ptr(flag), None, 0, it + 1, 0, 50_000, _lib.ptr(err), E.stream_ptr())
        eng.enqueue_encoder_forward(ws, run)
        _lib.call("arcvae_gate_set", _lib.ptr(flag), it + 1, 0, E.stream_ptr())
        for o in others:
            main.wait_stream(o)
    torch.cuda.synchronize()
    print(f"  (wall time of the 6 iterations: {time.perf_counter() - t_wall:.3f} s)")
    a = buf.cpu().numpy().reshape(cap, 2).astype(np.float64)[:nf] / 100.0
    stats(f"engine enc_fwd segment with {nblocked} other queue(s) spinning in a gate kernel (timeouts: {int(err.item())})", a - a[0, 0])
for nb_ in (1, 2, 1):
    gated_queue_test(nb_)
