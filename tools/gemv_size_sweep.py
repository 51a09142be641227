"""Time of ONE launch of the step's weight-streaming GEMV (csrc/gemv.hip, n = 11 rows, bf16 image, K = 4096, plain bf16 epilogue) against the bytes
it streams, N swept from 512 to 57 344 output rows (4 MB ... 470 MB; weights cycled through > 1.2 GB so nothing is re-read from a cache), and the
least-squares line  time = fixed + bytes / rate  through the measurements: the two terms DESIGN.md section 7 (round 5) prices every small kernel
of the step with.

    python tools/gemv_size_sweep.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from videollm_online_amd import _C

L = _C.lib()
torch.zeros(1, device="cuda")
K = 4096
pts = []
for N in (512, 1024, 2048, 4096, 6144, 8192, 12288, 16384, 28672, 40960, 57344):
    wb = N * K * 2
    us = C.c_double()
    _C.check(L.vlo_bench_gemv(N, K, 11, 1, 80, max(2, int(1.2e9 // wb) + 1), C.byref(us)))
    pts.append((wb, us.value))
    print(f"N={N:6d}: {wb / 1e6:7.1f} MB  {us.value:7.2f} us  {wb / us.value / 1e6:5.2f} TB/s over the launch", flush=True)
# least squares over the points from 16 MB up (below that the grid does not fill the chip)
xs, ys = zip(*[(b, t) for b, t in pts if b >= 16e6])
n = len(xs)
mx, my = sum(xs) / n, sum(ys) / n
slope = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
fixed = my - slope * mx
print(f"fit over >= 16 MB: time = {fixed:.2f} us + bytes / {1 / slope / 1e6:.2f} TB/s   (residuals: "
      + ", ".join(f"{y - (fixed + slope * x):+.2f}" for x, y in zip(xs, ys)) + " us)")
