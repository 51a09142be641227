#!/usr/bin/env bash
# mid-size batches: where the branch split and the ping-pong GEMM should start, and its tile height inside a branch
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c16
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for sp in 4 8; do for rows in 1152 2304; do for bm in 128 256; do
  echo "== split_min=$sp pp_min_rows=$rows bm=$bm"
  VLO_VIT_SPLIT_MIN=$sp VLO_VIT_PP_MIN_ROWS=$rows VLO_VIT_PP_BM=$bm timeout 200 python tools/probe_vit_b.py 4,6,8,10,12,14,16 20 2>&1 | grep "B=" | tee "$OUT/sweep_sp${sp}_rows${rows}_bm${bm}.txt"
done; done; done
exit 0
