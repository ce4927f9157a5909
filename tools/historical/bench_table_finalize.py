"""arcvae_table_finalize (one launch) vs the three launches it replaced (two split-K tile GEMMs + colsum), isolated."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mlx-vae_amd"))
import torch
from arcvae_hip import _lib
V, E, G = 80, 128, 1024
dT = torch.randn(V, G, device="cuda"); Wx0 = torch.randn(G, E, device="cuda"); emb = torch.randn(V, E, device="cuda")
dEmb = torch.zeros(V, E, device="cuda"); dWx0 = torch.zeros(G, E, device="cuda"); db0 = torch.zeros(G, device="cuda")
s = _lib.stream_ptr()
def fused():
    _lib.call("arcvae_table_finalize", _lib.ptr(dT), _lib.ptr(Wx0), E, _lib.ptr(emb), _lib.ptr(dEmb), _lib.ptr(dWx0), _lib.ptr(db0), V, E, G, s)
def three():
    _lib.gemm(False, False, V, E, G, dT, G, Wx0, E, dEmb, E, None, 1 | 4 | 8)
    _lib.gemm(True, False, G, E, V, dT, G, emb, E, dWx0, E, None, 1)
    _lib.call("arcvae_colsum_accum", _lib.ptr(dT), V, G, G, _lib.ptr(db0), 1.0, s)
fused(); torch.cuda.synchronize()
ref = (dT @ Wx0, dT.t() @ emb, dT.sum(0))
for a, b in zip((dEmb, dWx0, db0), ref):
    print("rel err", float((a - b).abs().max() / b.abs().max()))
for name, fn in (("fused", fused), ("three launches", three)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {10 * e0.elapsed_time(e1):.1f} us")
