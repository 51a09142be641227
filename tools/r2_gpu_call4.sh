#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2c4
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -s > "$OUT/gpu_suite.log" 2>&1; echo "gpu_suite exit $?" >> "$OUT/gpu_suite.log"
timeout 200 python tools/probe_vit_b.py 1,4,6,7,8,14,16 10 > "$OUT/vit_B_sweep.log" 2>&1
cd /tmp && export TMPDIR=/tmp
for B in 8 7; do
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_vit_b$B" -o vit_b$B -- python $ROOT/tools/probe_vit_b.py $B 10 > "$OUT/prof_vit_b$B.log" 2>&1
done
cd $ROOT
find "$OUT" -name "*.db" | head; find "$OUT" -name "*kernel_stats*" | head
for B in 8 7; do
  db=$(find "$OUT/prof_vit_b$B" -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$OUT/vit_b${B}_kernel_stats.csv"
done
tail -4 "$OUT/gpu_suite.log"; cat "$OUT/vit_B_sweep.log"; head -14 "$OUT/vit_b8_kernel_stats.csv" | cut -c1-160
# keep the merge small
find "$OUT" -name "*.db" -size +20M -delete
exit 0
