"""bf16-in / f32-accumulate tile GEMM (csrc/gemm.hip gemm_bf16_tile_kernel, flag ARCVAE_GEMM_BF16) against the exact-f32
MFMA tile kernel on the big shapes of BASELINE.json configs[2]: time per call (HIP events, graph replay), TFLOP/s, and
error against fp64 relative to sum |a||b|."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mlx-vae_amd"))
import torch
from arcvae_hip import _lib
dev = "cuda"
def run(tA, tB, M, N, K, flags, reps=10):
    A = torch.randn((K, M) if tA else (M, K), device=dev); B = torch.randn((N, K) if tB else (K, N), device=dev)
    C = torch.zeros(M, N, device=dev)
    f = lambda: _lib.gemm(tA, tB, M, N, K, A, A.shape[1], B, B.shape[1], C, N, None, flags)
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
    rows = slice(0, min(M, 512))
    Ad = (A.t() if tA else A)[rows].double(); Bd = (B.t() if tB else B).double()
    ref = Ad @ Bd
    mag = Ad.abs() @ Bd.abs()
    err = float(((C[rows].double() - ref).abs() / mag).max())
    return us, err
shapes = [(0, 1, 40960, 2048, 512), (0, 0, 40960, 512, 2048), (1, 0, 2048, 512, 40960), (0, 1, 40960, 80, 512),
          (0, 0, 40960, 512, 80), (0, 1, 5120, 1024, 256), (0, 0, 5120, 256, 1024), (1, 0, 1024, 256, 5120),
          (0, 1, 8192, 8192, 8192)]
for (tA, tB, M, N, K) in shapes:
    sk = _lib.GEMM_ACCUMULATE | _lib.GEMM_SPLITK if tA else 0
    us_b, e_b = run(tA, tB, M, N, K, sk | _lib.GEMM_BF16 | _lib.GEMM_NO_SKINNY)
    us_f, e_f = run(tA, tB, M, N, K, sk | _lib.GEMM_NO_SKINNY)
    fl = 2.0 * M * N * K
    print(f"{'T' if tA else 'N'}{'T' if tB else 'N'} [{M}x{N}] K={K}: bf16 {us_b:8.1f} us ({fl / us_b / 1e6:7.1f} TF, err {e_b:.1e})   "
          f"f32 {us_f:8.1f} us ({fl / us_f / 1e6:6.1f} TF, err {e_f:.1e})", flush=True)
