#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2c5
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python tools/probe_vit_b.py 1,4,7,8,14,16 10 > "$OUT/vit_B_sweep.log" 2>&1
timeout 200 python tools/bench_gemv.py 8b both > "$OUT/bench_gemv_8b.log" 2>&1
timeout 200 python tools/bench_gemv.py 70b-tp8 both > "$OUT/bench_gemv_70b_tp8.log" 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_driver_line.log" 2>&1
timeout 1500 python -m pytest tests -m gpu -q -s > "$OUT/gpu_suite.log" 2>&1; echo "gpu_suite exit $?" >> "$OUT/gpu_suite.log"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_vit_b8" -o vit_b8 -- python $ROOT/tools/probe_vit_b.py 8 10 > "$OUT/prof_vit_b8.log" 2>&1
cd $ROOT
db=$(find "$OUT/prof_vit_b8" -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$OUT/vit_b8_kernel_stats.csv"
find "$OUT" -name "*.db" -size +20M -delete
cat "$OUT/vit_B_sweep.log" "$OUT/bench_gemv_8b.log" "$OUT/bench_gemv_70b_tp8.log"; grep '^{' "$OUT/bench_driver_line.log" | cut -c1-400; tail -15 "$OUT/gpu_suite.log" | cut -c1-220; head -8 "$OUT/vit_b8_kernel_stats.csv" | cut -c1-150
exit 0
