"""GEMV micro-benchmark: the step's weight-streaming shapes (Llama-3-8B and Llama-3-70B, whole or one rank's shard at TP = 8), bf16 and fp8 e4m3
weight images, TB/s of weight bytes per shape (weights cycled through > 1 GB so the Infinity Cache cannot serve re-reads).

    python tools/bench_gemv.py [8b|70b|8b-tp8|70b-tp8] [bf16|fp8|both]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from videollm_online_amd import _C

L = _C.lib()
torch.zeros(1, device="cuda")
# epi: 5 rope, 4 resid, 3 swiglu(+norm), 0 partial, 1 bf16
SHAPES = {"8b": [("qkv+rope", 6144, 4096, 5), ("o+resid", 4096, 4096, 4), ("gate_up", 28672, 4096, 3), ("down/ks", 4096, 14336, 0),
                 ("lm_head", 128256, 4096, 1)],
          "70b": [("qkv+rope", 10240, 8192, 5), ("o+resid", 8192, 8192, 4), ("gate_up", 57344, 8192, 3), ("down/ks", 8192, 28672, 0),
                  ("lm_head", 128256, 8192, 1)],
          "8b-tp8": [("qkv+rope", 768, 4096, 5), ("o/ks", 4096, 512, 0), ("gate_up", 3584, 4096, 3), ("down/ks", 4096, 1792, 0),
                     ("lm_head", 16032, 4096, 1)],
          "70b-tp8": [("qkv+rope", 1280, 8192, 5), ("o/ks", 8192, 1024, 0), ("gate_up", 7168, 8192, 3), ("down/ks", 8192, 3584, 0),
                      ("lm_head", 16032, 8192, 1)]}
model = sys.argv[1] if len(sys.argv) > 1 else "8b"
which = sys.argv[2] if len(sys.argv) > 2 else "both"
FORCE_NBUF = int(os.environ.get("NBUF", "0"))
for fp8 in ([0, 1] if which == "both" else [1 if which == "fp8" else 0]):
    tot_us, tot_b = 0.0, 0.0
    for name, N, K, epi in SHAPES[model]:
        wb = N * K * (1 if fp8 else 2)
        nbuf = FORCE_NBUF or max(2, int(1.2e9 // wb) + 1)
        us = C.c_double()
        try:
            _C.check(L.vlo_bench_gemv(N, K, 11, epi | (0x100 if fp8 else 0), 60, nbuf, C.byref(us)))
        except RuntimeError as ex:
            print(f"{'fp8' if fp8 else 'bf16'} {name}: {ex}")
            continue
        print(f"{'fp8 ' if fp8 else 'bf16'} {name:9s} N={N:6d} K={K:5d}: {us.value:8.2f} us  {wb / 1e9 / (us.value * 1e-6) / 1e3:6.2f} TB/s", flush=True)
        if name != "lm_head":
            tot_us += us.value
            tot_b += wb
    print(f"{'fp8 ' if fp8 else 'bf16'} per-layer GEMV total {tot_us:.1f} us for {tot_b / 1e6:.1f} MB (ideal @6.3 TB/s: {tot_b / 6.3e12 * 1e6:.1f} us)")
