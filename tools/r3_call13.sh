#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c13
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for s in 3 4; do
  echo "== VLO_VIT_SMALL_STAGES=$s"
  VLO_VIT_SMALL_STAGES=$s timeout 200 python tools/probe_vit_b.py 1,2,3,4 20 2>&1 | grep "B=" | tee "$OUT/sweep_stages$s.txt"
done
cd /tmp && export TMPDIR=/tmp
VLO_VIT_SMALL_STAGES=3 timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/prof_b1_s3" -o vit -- python $ROOT/tools/probe_vit_b.py 1 10 > "$OUT/prof_b1_s3.log" 2>&1
db=$(find "$OUT/prof_b1_s3" -name "*.db" | head -1); [ -n "$db" ] && python $ROOT/tools/rocpd_stats.py "$db" > "$OUT/kernel_stats_vit_b1_s3.csv"
head -8 "$OUT/kernel_stats_vit_b1_s3.csv" | cut -c1-140
find "$OUT" -name "*.db" -delete
exit 0
