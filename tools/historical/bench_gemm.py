"""Time arcvae_gemm_f32 on the shapes the training step uses (isolated, back-to-back launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mlx-vae_amd"))
import torch
from arcvae_hip import _lib
shapes = [  # name, tA, tB, M, N, K, flags
    ("dWh  (TN, K=8128, split)", 1, 0, 1024, 256, 8128, 5),
    ("dWh chunk (K=2048, split)", 1, 0, 1024, 256, 2048, 5),
    ("dWh chunk (K=1216, split)", 1, 0, 1024, 256, 1216, 5),
    ("dec proj (NT, M=5120)", 0, 1, 5120, 1024, 256, 0),
    ("dec dh (NN, M=5120,K=1024)", 0, 0, 5120, 256, 1024, 0),
    ("dec dWx (TN, K=5120, split)", 1, 0, 1024, 256, 5120, 5),
    ("fc_out (NT, N=80)", 0, 1, 5120, 80, 256, 0),
    ("dWout (TN, M=80, split)", 1, 0, 80, 256, 5120, 5),
    ("dh=dL.Wout (NN,K=80)", 0, 0, 5120, 256, 80, 0),
    ("dEmb (skinny NN, M=80,K=1024)", 0, 0, 80, 128, 1024, 1),
    ("dWx0 (TN, K=80)", 1, 0, 1024, 128, 80, 1),
    ("heads lh (skinny NT 64x512x512)", 0, 1, 64, 512, 512, 2),
    ("heads dcomb (skinny NN 64x512x512)", 0, 0, 64, 512, 512, 1),
]


if len(sys.argv) > 1 and sys.argv[1] == "big":  # BASELINE.json configs[2]: H512 L4 bs 512 (B*V = 40960 decoder rows)
    shapes = [
        ("big dWh chunk (TN, K=19968, split)", 1, 0, 2048, 512, 19968, 5),
        ("big dWh chunk (TN, K=9728, split)", 1, 0, 2048, 512, 9728, 5),
        ("big dec proj (NT, M=40960)", 0, 1, 40960, 2048, 512, 0),
        ("big dec dh (NN, M=40960,K=2048)", 0, 0, 40960, 512, 2048, 0),
        ("big dec dWx (TN, K=40960, split)", 1, 0, 2048, 512, 40960, 5),
        ("big fc_out (NT, N=80)", 0, 1, 40960, 80, 512, 0),
    ]


def time_one(tA, tB, M, N, K, flags):
    A = torch.randn((K, M) if tA else (M, K), device="cuda")
    B = torch.randn((N, K) if tB else (K, N), device="cuda")
    Cm = torch.zeros(M, N, device="cuda")
    for _ in range(3):
        _lib.gemm(bool(tA), bool(tB), M, N, K, A, A.shape[1], B, B.shape[1], Cm, N, None, flags)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50 if M * N * K < 4e10 else 10
    e0.record()
    for _ in range(n):
        _lib.gemm(bool(tA), bool(tB), M, N, K, A, A.shape[1], B, B.shape[1], Cm, N, None, flags)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for name, tA, tB, M, N, K, flags0 in shapes:
    for tile, tf in (("auto", 0), ("t64", 16), ("t128", 32)):
        skinny = M <= 256 and not tA and K % 64 == 0
        if tf and skinny:
            continue
        us = time_one(tA, tB, M, N, K, flags0 | tf)
        print(f"{name:38s} {tile:5s} {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s")
