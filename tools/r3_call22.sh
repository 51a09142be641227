#!/usr/bin/env bash
# configs[4] with BOTH halves on one GPU: Llama-3-70B shape, fp8 weights, TP = 1 + SigLIP-so400m/14-384; per-kernel view of the so400m tower
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c22
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --model llama-3-70b --weight-dtype fp8 --vit siglip-so400m14-384 --no-cpu-baseline > "$OUT/bench_70b_fp8_so400m.json" 2> "$OUT/bench_70b.err"; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3c22/bench_70b_fp8_so400m.json").read().strip().splitlines()[-1])
print(d["value"], d["p50_frame_latency_ms"], d["p95_frame_latency_ms"], d["encode_stage"], d["full_stream"]["frames_per_s"], d["roofline"]["frac"], d["config"]["workload"][:80])
PY
cd /tmp && export TMPDIR=/tmp
VLO_PROBE_VIT=so400m timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_so400m_b28" -o vit -- python $ROOT/tools/probe_vit_b.py 28 6 > "$OUT/prof_so400m_b28.log" 2>&1
db=$(find "$OUT/prof_so400m_b28" -name "*.db" | head -1); [ -n "$db" ] && python $ROOT/tools/rocpd_stats.py "$db" > "$OUT/kernel_stats_so400m_b28.csv"
head -12 "$OUT/kernel_stats_so400m_b28.csv" | cut -c1-150
find "$OUT" -name "*.db" -delete
exit 0
