#!/usr/bin/env bash
# GELU with the pinned fp32 product: fc1 speed and bit-identity across the GEMM kernels; the tower at the usual batch sizes
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c17
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
GEMM_PROBE_GEMM=fc1 timeout 200 tools/_bin/gemm_probe 8 16 28 > "$OUT/gemm_probe_fc1.txt" 2>&1
grep -v "cb[1248] " "$OUT/gemm_probe_fc1.txt" | cut -c1-120
timeout 200 python tools/probe_vit_b.py 1,8,16,28,56 20 2>&1 | grep "B=" | tee "$OUT/sweep.txt"
exit 0
