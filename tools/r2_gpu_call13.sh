#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2c13
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python -m pytest tests/test_gpu_vit.py tests/test_gpu_liveinfer.py -m gpu -q -s > "$OUT/vit_tests.log" 2>&1; echo "exit $?" >> "$OUT/vit_tests.log"
timeout 200 python tools/probe_vit_b.py 4,8,12,14,16,24,32 10 > "$OUT/vit_split.log" 2>&1
VLO_VIT_SPLIT_MIN=0 timeout 200 python tools/probe_vit_b.py 8,14,16,32 10 > "$OUT/vit_nosplit.log" 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --prefetch-frames 16 > "$OUT/bench_pf16.log" 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --prefetch-frames 14 > "$OUT/bench_pf14.log" 2>&1
echo split; grep "B=" "$OUT/vit_split.log"; echo nosplit; grep "B=" "$OUT/vit_nosplit.log"
for f in bench_pf16 bench_pf14; do grep '^{' "$OUT/$f.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', d['value'], 'fps p50', d['p50_frame_latency_ms'], 'enc', d['encode_stage'])"; done
grep -a "two-branch" "$OUT/vit_tests.log"; tail -3 "$OUT/vit_tests.log"
exit 0
