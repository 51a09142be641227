#!/usr/bin/env bash
# mid-size batches: branch split from 4 frames, ping-pong GEMM from 2304 rows, tile height by the count of 256-row tiles
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c18
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
BS=4,5,6,7,8,10,12,14,16,20,24,28,40,56
echo "== shipped defaults"; timeout 200 python tools/probe_vit_b.py $BS 20 2>&1 | grep "B=" | tee "$OUT/sweep_default.txt"
for T in 0 48 96 128 192; do
  echo "== split 4, rows 2304, bm256 from $T tiles (0 = fill rule)"
  VLO_VIT_SPLIT_MIN=4 VLO_VIT_PP_MIN_ROWS=2304 VLO_VIT_PP_BM256_MIN_TILES=$T timeout 200 python tools/probe_vit_b.py $BS 20 2>&1 | grep "B=" | tee "$OUT/sweep_T$T.txt"
done
echo "== split 4, rows 1728, bm256 from 96 tiles"
VLO_VIT_SPLIT_MIN=4 VLO_VIT_PP_MIN_ROWS=1728 VLO_VIT_PP_BM256_MIN_TILES=96 timeout 200 python tools/probe_vit_b.py $BS 20 2>&1 | grep "B=" | tee "$OUT/sweep_r1728_T96.txt"
exit 0
