#!/usr/bin/env bash
# Round-3 GPU call 5: attention probe, full GPU suite, bench driver line.  Results: gpurun_out/r3c5/.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/${R3OUT:-r3c5}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 tools/_bin/attn_probe 8 14 16 28 32 > "$OUT/attn_probe.txt" 2>&1
timeout 300 python tools/probe_vit_b.py 1,8,16,28,56 10 > "$OUT/vit_sweep_default.txt" 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x > "$OUT/gpu_suite.log" 2>&1; echo "gpu_suite exit $?" >> "$OUT/gpu_suite.log"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_line.json" 2> "$OUT/bench_driver_line.err"
cat "$OUT/attn_probe.txt"; grep "B=" "$OUT/vit_sweep_default.txt"; tail -4 "$OUT/gpu_suite.log" | cut -c1-200
python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_driver_line.json").read().strip().splitlines()[-1])
    print("bench", d["value"], "fps p50", d["p50_frame_latency_ms"], "p95", d["p95_frame_latency_ms"], "enc", d["encode_stage"], "full", d.get("full_stream",{}).get("frames_per_s"), "hbm", d["stream_hbm_roofline"]["frac_of_hbm_peak"], "roof", d["roofline"]["frac"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as ex:
    print("bench FAILED", ex); print(open("$OUT/bench_driver_line.err").read()[-1500:])
PY
exit 0
