#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2c9
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
VLO_FIXUP=0 timeout 200 python tools/probe_step.py > "$OUT/step_fixup0.log" 2>&1
VLO_FIXUP=1 timeout 200 python tools/probe_step.py > "$OUT/step_fixup1.log" 2>&1
timeout 200 python tools/probe_step.py --weight-dtype fp8 > "$OUT/step_fp8.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_llm.py tests/test_gpu_liveinfer.py tests/test_gpu_eval.py tests/test_gpu_fp8.py tests/test_reference_liveinfer_flow.py tests/test_gpu_vit.py -m gpu -q -s > "$OUT/gpu_subset.log" 2>&1; echo "subset exit $?" >> "$OUT/gpu_subset.log"
VLO_FIXUP=0 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_k20_fixup0.log" 2>&1
VLO_FIXUP=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_k20_fixup1.log" 2>&1
timeout 200 python tools/probe_vit_b.py 1,8,14 10 > "$OUT/vit_sweep.log" 2>&1
cat "$OUT/step_fixup0.log" "$OUT/step_fixup1.log" "$OUT/step_fp8.log" | grep "Lc~"
for f in bench_k20_fixup0 bench_k20_fixup1; do grep '^{' "$OUT/$f.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', d['value'], 'fps p50', d['p50_frame_latency_ms'], 'p95', d['p95_frame_latency_ms'], 'full', d.get('full_stream',{}).get('frames_per_s'), 'hbm', d['stream_hbm_roofline']['frac_of_hbm_peak'], 'full hbm', d.get('full_stream',{}).get('frac_of_hbm_peak'), 'roof', d['roofline']['frac'])"; done
grep "B=" "$OUT/vit_sweep.log"; grep -a "Lc=13245\|Lc=4096\|full\]" "$OUT/gpu_subset.log" | cut -c1-220 | head; tail -8 "$OUT/gpu_subset.log" | cut -c1-200
exit 0
