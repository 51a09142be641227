#!/usr/bin/env bash
# (1) one frame with the final split-K defaults  (2) mid-size batches: branch split and tile height of the ping-pong GEMM
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c15
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 500 python -m pytest tests/test_gpu_vit.py -x -q > "$OUT/pytest_vit.log" 2>&1; echo "pytest vit exit $?"
tail -2 "$OUT/pytest_vit.log"
echo "== defaults"; timeout 200 python tools/probe_vit_b.py 1,2,3,4,6,8,12,16,20,24,28 20 2>&1 | grep "B=" | tee "$OUT/sweep_default.txt"
echo "== no branch split (VLO_VIT_SPLIT_MIN=0)"; VLO_VIT_SPLIT_MIN=0 timeout 200 python tools/probe_vit_b.py 8,12,16,20,24,28 20 2>&1 | grep "B=" | tee "$OUT/sweep_nosplit.txt"
for bm in 128 256; do
  echo "== VLO_VIT_PP_BM=$bm (branches as default)"; VLO_VIT_PP_BM=$bm timeout 200 python tools/probe_vit_b.py 8,12,16,20,24,28 20 2>&1 | grep "B=" | tee "$OUT/sweep_bm$bm.txt"
  echo "== VLO_VIT_PP_BM=$bm no split"; VLO_VIT_SPLIT_MIN=0 VLO_VIT_PP_BM=$bm timeout 200 python tools/probe_vit_b.py 8,12,16,20,24,28 20 2>&1 | grep "B=" | tee "$OUT/sweep_bm${bm}_nosplit.txt"
done
echo "== pp from 2304 rows in a branch (VLO_VIT_PP_MIN_ROWS=2304) + split from 8"; VLO_VIT_PP_MIN_ROWS=2304 VLO_VIT_SPLIT_MIN=8 timeout 200 python tools/probe_vit_b.py 8,12,16 20 2>&1 | grep "B=" | tee "$OUT/sweep_split8.txt"
exit 0
