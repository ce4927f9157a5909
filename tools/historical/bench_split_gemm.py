"""Split-bf16 TN GEMM (csrc/gemm.hip) against the exact-f32 MFMA tile kernel on the weight-gradient shapes of the
default training step: time per call (HIP events, graph replay) and error against fp64."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mlx-vae_amd"))
import numpy as np, torch
from arcvae_hip import _lib
dev = "cuda"
def run(M, N, K, flags, reps=20):
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); C = torch.zeros(M, N, device=dev)
    f = lambda: _lib.gemm(True, False, M, N, K, A, M, B, N, C, N, None, flags)
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    C.zero_(); f(); torch.cuda.synchronize()
    ref = A.double().t() @ B.double()
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    return us, err
for (M, N, K) in [(1024, 256, 3072), (1024, 256, 5120), (1024, 256, 8192), (80, 1024, 3072), (80, 256, 5120), (2048, 512, 8192)]:
    us_s, e_s = run(M, N, K, _lib.GEMM_ACCUMULATE | _lib.GEMM_SPLITK)
    os.environ["X"] = "1"
    us_f, e_f = run(M, N, K, _lib.GEMM_ACCUMULATE | _lib.GEMM_SPLITK | 16)   # TILE64 flag forces the f32 tile kernel (split-K)
    fl = 2.0 * M * N * K
    print(f"[{M}x{N}] K={K}: split-bf16 {us_s:7.1f} us ({fl / us_s / 1e6:6.1f} TF-equiv, err {e_s:.1e})   f32 MFMA {us_f:7.1f} us ({fl / us_f / 1e6:6.1f} TF, err {e_f:.1e})")
