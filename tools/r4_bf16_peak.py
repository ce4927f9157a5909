"""What the vendor library's bf16 GEMM reaches on THIS box on random data (hipBLASLt through torch.matmul): the practical ceiling of
the bf16 matrix pipe that DESIGN.md §6i sets beside the nominal 2.5 PFLOP/s.  python tools/r4_bf16_peak.py"""
import torch
dev = torch.device("cuda", 0)
def run(M, N, K, zeros=False):
    a = (torch.zeros if zeros else torch.randn)(M, K, device=dev, dtype=torch.bfloat16)
    b = (torch.zeros if zeros else torch.randn)(K, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"bf16 {M} x {N} x {K} {'zeros ' if zeros else 'random'}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.0f} TFLOP/s")
for shape in ((8192, 8192, 8192), (4096, 4096, 16384), (2048, 512, 16384), (2048, 3584, 16384)):
    run(*shape)
run(8192, 8192, 8192, zeros=True)
