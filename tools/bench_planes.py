"""The operand-plane weight-gradient GEMM (gemm.hip: wgrad_planes_kernel) alone: python tools/bench_planes.py [B H L T nT]
One call = the stack's dWh_l / dWx_l over nT time steps (arcvae_enc_lstm_wgrad parts = GEMMs | planes), random planes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mlx-vae_amd"))
import torch
from arcvae_hip import _lib
from arcvae_hip._lib import ptr, ptr_array, stream_ptr
B, H, L, T, nT = (int(a) for a in (sys.argv[1:6] + ["512", "512", "4", "128", "32"][len(sys.argv) - 1:]))
lib = _lib.load()
dev = torch.device("cuda", 0)
G, V, E = 4 * H, 80, 128
f32 = dict(dtype=torch.float32, device=dev)
hseq = torch.randn(L, T, B, H, **f32); dG = torch.randn(L, T, B, G, **f32)
hpl = torch.randn(L * T * B * H * 3 // 2, **f32).view(torch.bfloat16).normal_().view(torch.float32)
gpl = torch.randn(L * T * B * G * 3 // 2, **f32).view(torch.bfloat16).normal_().view(torch.float32)
x_tb = torch.zeros(T, B, dtype=torch.int32, device=dev)
emb = torch.randn(V, E, **f32); wx0 = torch.randn(G, E, **f32)
dtab = torch.zeros(V, G, **f32); onehot = torch.zeros(T * B, V, **f32); demb = torch.zeros(V, E, **f32)
dwx = [torch.zeros(G, E if l == 0 else H, **f32) for l in range(L)]
dwh = [torch.zeros(G, H, **f32) for l in range(L)]
dbs = [torch.zeros(G, **f32) for l in range(L)]
pwx, _a = ptr_array(dwx); pwh, _b = ptr_array(dwh); pbs, _c = ptr_array(dbs)
def call(parts, t_lo, t_hi):
    rc = lib.arcvae_enc_lstm_wgrad(ptr(x_tb), ptr(emb), ptr(wx0), ptr(hseq), ptr(dG), ptr(dtab), ptr(onehot), ptr(demb), pwx, pwh, pbs,
                                   B, T, V, E, H, L, t_lo, t_hi, 0, 0, parts, ptr(hpl), ptr(gpl),
                                   (C.c_long * 4)(hpl.numel(), gpl.numel(), 0, 0), stream_ptr())
    assert rc == 0, rc
flops = 2.0 * nT * B * G * H * (2 * L - 1)
for name, parts in (("planes", 1 | 16 | 2048), ("three-piece tile GEMM (f32 sources)", 1 | 16 | 1024)):
    for _ in range(2):
        call(parts, 32, 32 + nT)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        call(parts, 32, 32 + nT)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name}: {ms * 1e3:.0f} us per call (GEMMs + bias column sums), {flops / ms / 1e9:.0f} TFLOP/s f32-equivalent")
