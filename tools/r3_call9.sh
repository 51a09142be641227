#!/usr/bin/env bash
# small-batch ViT: direct-to-LDS small tiles (VLO_VIT_SMALL_STAGES) against the register ring
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c9
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 500 python -m pytest tests/test_gpu_vit.py -x -q > "$OUT/pytest_vit.log" 2>&1; echo "pytest vit exit $?" | tee -a "$OUT/pytest_vit.log"
tail -3 "$OUT/pytest_vit.log"
for s in 0 4 6; do
  echo "== VLO_VIT_SMALL_STAGES=$s"
  VLO_VIT_SMALL_STAGES=$s timeout 200 python tools/probe_vit_b.py 1,2,4,6,8 20 2>&1 | grep "B=" | tee "$OUT/sweep_stages$s.txt"
done
cd /tmp && export TMPDIR=/tmp
for s in 4 6; do
  VLO_VIT_SMALL_STAGES=$s timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/prof_b1_s$s" -o vit -- python $ROOT/tools/probe_vit_b.py 1 10 > "$OUT/prof_b1_s$s.log" 2>&1
  db=$(find "$OUT/prof_b1_s$s" -name "*.db" | head -1); [ -n "$db" ] && python $ROOT/tools/rocpd_stats.py "$db" > "$OUT/kernel_stats_vit_b1_s$s.csv"
  echo "== kernel stats B=1 stages $s"; head -16 "$OUT/kernel_stats_vit_b1_s$s.csv" | cut -c1-140
done
find "$OUT" -name "*.db" -delete
exit 0
