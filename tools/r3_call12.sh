#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c12
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for s in 200 300 450 700 100000; do
  echo "== VLO_VIT_BIG_TILES=$s"
  VLO_VIT_BIG_TILES=$s timeout 200 python tools/probe_vit_b.py 3,4,5,6,7 20 2>&1 | grep "B=" | tee "$OUT/sweep_big$s.txt"
done
echo "== pp from fewer rows (VLO_VIT_PP_MIN_ROWS)"
for r in 2048 3072; do
  echo "-- min rows $r"
  VLO_VIT_PP_MIN_ROWS=$r timeout 200 python tools/probe_vit_b.py 4,5,6,7 20 2>&1 | grep "B=" | tee "$OUT/sweep_pprows$r.txt"
done
exit 0
