#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c8
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
for B in 1 4 8; do
  timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/prof_vit_b$B" -o vit -- python $ROOT/tools/probe_vit_b.py $B 10 > "$OUT/prof_vit_b$B.log" 2>&1
  db=$(find "$OUT/prof_vit_b$B" -name "*.db" | head -1); [ -n "$db" ] && python $ROOT/tools/rocpd_stats.py "$db" > "$OUT/kernel_stats_vit_b$B.csv"
done
find "$OUT" -name "*.db" -delete
for B in 1 4 8; do echo "== B=$B"; grep "B=" "$OUT/prof_vit_b$B.log"; head -14 "$OUT/kernel_stats_vit_b$B.csv" | cut -c1-130; done
exit 0
