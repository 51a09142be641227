#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r2prio
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { name=$1; shift; timeout 300 python bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; python - <<PY
import json
try:
    d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], "p50", d["p50_frame_latency_ms"], "p95", d["p95_frame_latency_ms"], "full", d["full_stream"]["frames_per_s"], "fullp50", d["full_stream"]["p50_frame_latency_ms"])
except Exception as ex:
    print("$name FAILED", ex)
PY
}
run default
run hiprio --llm-high-priority
run hiprio_serial --llm-high-priority --encode-on-main-stream
run hiprio_pf32 --llm-high-priority --prefetch-frames 32
run hiprio_serial_pf8 --llm-high-priority --encode-on-main-stream --prefetch-frames 8
grep -h "priority" "$OUT/hiprio.err" | head -2
exit 0
