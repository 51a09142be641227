#!/usr/bin/env bash
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r2fp8; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_fp8.py -q 2>&1 | tail -3
timeout 120 python tools/bench_gemv.py 8b fp8 2>&1 | grep -v amdgpu.ids
timeout 120 python tools/bench_gemv.py 70b-tp8 fp8 2>&1 | grep -v amdgpu.ids
timeout 200 python tools/probe_step.py --weight-dtype fp8 --iters 30 --lens 0,15360 2>&1 | grep "Lc~"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --weight-dtype fp8 --no-cpu-baseline > $OUT/bench_fp8_k20.json 2> $OUT/bench_fp8.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_fp8_k20.json").read().strip().splitlines()[-1])
print("bench fp8", d["value"], "p50", d["p50_frame_latency_ms"], "p95", d["p95_frame_latency_ms"], "full", d["full_stream"]["frames_per_s"], "roof", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
PY
exit 0
