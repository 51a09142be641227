"""GEMV micro-benchmark: the Llama-3-8B step shapes, GB/s per shape (weights cycled through >1 GB)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videollm_online_amd import _C
L = _C.lib()
torch.zeros(1, device="cuda")
SHAPES = [("qkv+rope", 6144, 4096, 5), ("o+resid", 4096, 4096, 4), ("gate_up", 28672, 4096, 3), ("down/ks4", 4096, 14336, 0), ("lm_head", 128256, 4096, 1)]   # epi: 5 rope, 4 resid, 3 swiglu(+norm), 0 partial, 1 bf16
tot_us = 0
FORCE_NBUF = int(os.environ.get("NBUF", "0"))
for name, N, K, epi in SHAPES:
    nbuf = FORCE_NBUF or max(2, int(1.2e9 // (N * K * 2)) + 1)
    us = C.c_double()
    _C.check(L.vlo_bench_gemv(N, K, 11, epi, 60, nbuf, C.byref(us)))
    gb = N * K * 2 / 1e9
    print(f"{name:8s} N={N:6d} K={K:5d}: {us.value:8.2f} us  {gb / (us.value * 1e-6) / 1e3:6.2f} TB/s", flush=True)
    if name != "lm_head": tot_us += us.value
print(f"per-layer GEMV total {tot_us:.1f} us (ideal @6.3TB/s: {436.2e6/6.3e12*1e6:.1f} us)")
