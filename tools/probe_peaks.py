"""Measured machine ceilings on this MI355X (SURVEY.md §8d asks for them next to the spec numbers): device-to-device copy
and read-only streaming bandwidth, library bf16/fp16 GEMM rate.  torch / hipBLASLt kernels only — calibration, not product."""

import torch


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    n = 2 * 1024 ** 3                                         # 4 GiB of bf16 per buffer
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_()
    y = torch.empty_like(x)
    t = timed(lambda: y.copy_(x), 10)
    print(f"D2D copy 4 GiB: {t*1e3:.2f} ms -> {2 * x.numel() * 2 / t / 1e12:.2f} TB/s (read + write), {x.numel() * 2 / t / 1e12:.2f} TB/s each way")
    xi = x.view(torch.int16)
    t = timed(lambda: xi.max(), 10)
    print(f"read-only reduction over 4 GiB: {t*1e3:.2f} ms -> {x.numel() * 2 / t / 1e12:.2f} TB/s")
    for dt in (torch.bfloat16, torch.float16):
        M = 8192
        a = torch.randn(M, M, device="cuda", dtype=dt)
        b = torch.randn(M, M, device="cuda", dtype=dt)
        t = timed(lambda: torch.matmul(a, b), 10)
        print(f"hipBLASLt {dt} {M}^3 GEMM: {t*1e3:.2f} ms -> {2 * M ** 3 / t / 1e12:.0f} TFLOP/s")


if __name__ == "__main__":
    main()
