#!/usr/bin/env bash
# so400m attention: 32-query blocks (VLO_VIT_ATTN_QS=2) against 64-query blocks
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c23
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for qs in 4 2; do
  echo "== VLO_VIT_ATTN_QS=$qs"
  VLO_VIT_ATTN_QS=$qs VLO_PROBE_VIT=so400m timeout 300 python tools/probe_vit_b.py 1,4,16,28,56 10 2>&1 | grep "B=\|rror" | tee "$OUT/so400m_qs$qs.txt"
done
VLO_VIT_ATTN_QS=2 timeout 600 python -m pytest tests/test_gpu_vit.py -x -q -k "so400m" 2>&1 | tail -2
exit 0
