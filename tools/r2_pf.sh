#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r2pf
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for pf in 0 2 1; do
  VLO_GEMV_PF=$pf timeout 200 python tools/probe_step.py --iters 30 --lens 0,4096,15360 > "$OUT/step_pf$pf.txt" 2>&1
done
grep -h "Lc~" "$OUT"/step_pf*.txt
for pf in 0 2; do
VLO_GEMV_PF=$pf timeout 300 python bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline > "$OUT/bench_pf$pf.json" 2> "$OUT/bench_pf$pf.err"
python - <<PY
import json
d=json.loads(open("$OUT/bench_pf$pf.json").read().strip().splitlines()[-1])
print("bench pf$pf", d["value"], "p50", d["p50_frame_latency_ms"], "full", d["full_stream"]["frames_per_s"], "roof", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
PY
done
timeout 600 python -m pytest tests/test_gpu_llm.py -q -k "stream_parity or golden or full_depth or config2" > "$OUT/gpu_llm.log" 2>&1; tail -3 "$OUT/gpu_llm.log"
exit 0
