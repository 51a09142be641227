#!/usr/bin/env bash
# small-batch ViT: split-K of the residual GEMMs (VLO_VIT_SPLITK = cap on the K slices; 1 = off)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c14
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 500 python -m pytest tests/test_gpu_vit.py -x -q > "$OUT/pytest_vit.log" 2>&1; echo "pytest vit exit $?"
tail -3 "$OUT/pytest_vit.log"
for s in 1 2 4; do
  echo "== VLO_VIT_SPLITK=$s"
  VLO_VIT_SPLITK=$s timeout 200 python tools/probe_vit_b.py 1,2,3,4 20 2>&1 | grep "B=" | tee "$OUT/sweep_splitk$s.txt"
done
echo "== VLO_VIT_SPLITK=4 VLO_VIT_SPLITK_MIN_TILES=4"
VLO_VIT_SPLITK=4 VLO_VIT_SPLITK_MIN_TILES=4 timeout 200 python tools/probe_vit_b.py 1,2,3,4 20 2>&1 | grep "B=" | tee "$OUT/sweep_splitk4_min4.txt"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/prof_b1" -o vit -- python $ROOT/tools/probe_vit_b.py 1 10 > "$OUT/prof_b1.log" 2>&1
db=$(find "$OUT/prof_b1" -name "*.db" | head -1); [ -n "$db" ] && python $ROOT/tools/rocpd_stats.py "$db" > "$OUT/kernel_stats_vit_b1.csv"
head -10 "$OUT/kernel_stats_vit_b1.csv" | cut -c1-150
find "$OUT" -name "*.db" -delete
exit 0
