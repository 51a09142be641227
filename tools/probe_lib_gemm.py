"""What the ROCm libraries (hipBLASLt / rocBLAS through torch.matmul) reach on the SigLIP-L GEMM shapes, fp16 in / fp32 accumulate, next to
the engine's own vit_gemm_kernel numbers (profiles/): M = 576 B rows; qkv N 3072 K 1024, out-proj N 1024 K 1024, fc1 N 4096 K 1024,
fc2 N 1024 K 4096.    python tools/probe_lib_gemm.py [B ...]"""
import sys
import torch

def bench(M, N, K, iters=50):
    a = torch.randn(M, K, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16)
    for _ in range(5):
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(a, w.t())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    return us, 2.0 * M * N * K / us / 1e6

for B in [int(v) for v in sys.argv[1:]] or [8, 16, 32]:
    M = 576 * B
    for name, N, K in (("qkv", 3072, 1024), ("out", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
        us, tf = bench(M, N, K)
        print(f"B={B:3d} {name}: M {M} N {N} K {K}: {us:7.1f} us  {tf:6.0f} TFLOP/s", flush=True)
us, tf = bench(8192, 8192, 8192, 10)
print(f"8192^3: {us:.0f} us {tf:.0f} TFLOP/s")
