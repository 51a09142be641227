#!/usr/bin/env bash
# A/B of the column-packed attention kernel against the chunk kernel (one GPU call).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2attn
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_llm.py -q -s > "$OUT/gpu_llm.log" 2>&1; echo "exit $?" >> "$OUT/gpu_llm.log"
LENS=0,4096,15360,61440
VLO_ATTN_COLS=0 timeout 200 python tools/probe_step.py --iters 20 --lens $LENS > "$OUT/step_chunk.txt" 2>&1
timeout 200 python tools/probe_step.py --iters 20 --lens $LENS > "$OUT/step_cols.txt" 2>&1
VLO_ATTN_BLOCKS=512 timeout 200 python tools/probe_step.py --iters 20 --lens $LENS > "$OUT/step_cols_b512.txt" 2>&1
VLO_ATTN_BLOCKS=128 timeout 200 python tools/probe_step.py --iters 20 --lens $LENS > "$OUT/step_cols_b128.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_cols" -o p -- python $ROOT/tools/probe_step.py --iters 10 --lens 15360 > "$OUT/prof_cols.log" 2>&1
VLO_ATTN_COLS=0 timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_chunk" -o p -- python $ROOT/tools/probe_step.py --iters 10 --lens 15360 > "$OUT/prof_chunk.log" 2>&1
cd $ROOT
for v in cols chunk; do db=$(find "$OUT/prof_$v" -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$OUT/kernel_stats_$v.csv"; done
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_k20.json" 2> "$OUT/bench.err"
grep -h "Lc~" "$OUT"/step_*.txt; grep -h "attn" "$OUT"/kernel_stats_*.csv | cut -c1-160
python - <<PY
import json
d=json.loads(open("$OUT/bench_k20.json").read().strip().splitlines()[-1])
print("bench", d["value"], "p50", d["p50_frame_latency_ms"], "full", d["full_stream"]["frames_per_s"], "hbm", d["stream_hbm_roofline"]["frac_of_hbm_peak"])
PY
grep -E "passed|failed|attention Lc|Error" "$OUT/gpu_llm.log" | cut -c1-200 | tail -12
exit 0
