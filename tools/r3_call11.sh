#!/usr/bin/env bash
# small-batch ViT: where the 128x128 / 8-wave tile takes over from the 64x64 one (VLO_VIT_BIG_TILES = smallest count of 128x128 tiles)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c11
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for s in 1 40 100 150 200; do
  echo "== VLO_VIT_BIG_TILES=$s"
  VLO_VIT_BIG_TILES=$s timeout 200 python tools/probe_vit_b.py 1,2,3,4,6 20 2>&1 | grep "B=" | tee "$OUT/sweep_big$s.txt"
done
cd /tmp && export TMPDIR=/tmp
for s in 1; do
  VLO_VIT_BIG_TILES=$s timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/prof_b1_big$s" -o vit -- python $ROOT/tools/probe_vit_b.py 1 10 > "$OUT/prof_b1_big$s.log" 2>&1
  db=$(find "$OUT/prof_b1_big$s" -name "*.db" | head -1); [ -n "$db" ] && python $ROOT/tools/rocpd_stats.py "$db" > "$OUT/kernel_stats_vit_b1_big$s.csv"
  echo "== kernel stats B=1 big $s"; head -10 "$OUT/kernel_stats_vit_b1_big$s.csv" | cut -c1-140
done
find "$OUT" -name "*.db" -delete
exit 0
