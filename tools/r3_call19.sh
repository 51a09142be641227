#!/usr/bin/env bash
# new small / mid batch defaults: ViT GPU suite, the batch sweep, the driver's bench line
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c19
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_vit.py -x -q -s > "$OUT/pytest_vit.log" 2>&1; echo "pytest vit exit $?"
tail -2 "$OUT/pytest_vit.log"; grep "two-branch\|256-tile" "$OUT/pytest_vit.log" | cut -c1-150
timeout 200 python tools/probe_vit_b.py 1,2,3,4,5,6,7,8,10,12,14,16,20,24,28,32,40,56 20 2>&1 | grep "B=" | tee "$OUT/vit_batch_sweep.txt"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_k20.json" 2> "$OUT/bench_k20.err"; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3c19/bench_k20.json").read().strip().splitlines()[-1])
print(d["value"], d["p50_frame_latency_ms"], d["p95_frame_latency_ms"], d["encode_stage"], d["full_stream"]["frames_per_s"], d["roofline"]["frac"])
PY
exit 0
