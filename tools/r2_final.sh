#!/usr/bin/env bash
# Final round-2 check (one GPU call): the GPU suite on the final sources, the driver's bench line (bf16 and fp8), the encode sweep.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2final
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1300 python -m pytest tests -m gpu -q > "$OUT/gpu_suite.log" 2>&1; echo "gpu_suite exit $?" >> "$OUT/gpu_suite.log"
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_line_k20.json" 2> "$OUT/bench_driver_line.err"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --weight-dtype fp8 --no-cpu-baseline > "$OUT/bench_fp8_k20.json" 2> "$OUT/bench_fp8.err"
timeout 200 python tools/probe_vit_b.py 1,8,12,16,32 10 > "$OUT/vit_batch_sweep.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_vit_b16" -o vit -- python $ROOT/tools/probe_vit_b.py 16 10 > "$OUT/prof_vit_b16.log" 2>&1
cd $ROOT
db=$(find "$OUT/prof_vit_b16" -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$OUT/kernel_stats_vit_b16.csv"
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
for f in bench_driver_line_k20 bench_fp8_k20; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    print("$f", d["value"], "fps p50", d["p50_frame_latency_ms"], "p95", d["p95_frame_latency_ms"], "enc", d["encode_stage"]["frac_of_mfma_peak"], "full", d.get("full_stream",{}).get("frames_per_s"), "hbm", d["stream_hbm_roofline"]["frac_of_hbm_peak"], "roof", d["roofline"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as ex:
    print("$f", "FAILED", ex)
PY
done
grep "B=" "$OUT/vit_batch_sweep.txt"; head -12 "$OUT/kernel_stats_vit_b16.csv" | cut -c1-150; tail -6 "$OUT/gpu_suite.log" | cut -c1-220
exit 0
