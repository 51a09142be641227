#!/usr/bin/env bash
# Round-3 GPU call 7: fp8 expansion pipelining A/B, long-context parity (66 k, TP = 8 at 13 k), configs[3] rehearsal, 70B fp8 line, CPU windows
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/${R3OUT:-r3c7}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for pipe in 0 1 0 1; do VLO_FP8_PIPE=$pipe timeout 120 python tools/bench_gemv.py 8b fp8 2>&1 | grep -v amdgpu | sed "s/^/pipe=$pipe /" >> "$OUT/fp8_pipe_ab.txt"; done
VLO_FP8_PIPE=1 timeout 300 python -m pytest tests/test_gpu_fp8.py -m gpu -q -x > "$OUT/test_fp8_pipe.log" 2>&1; echo "exit $?" >> "$OUT/test_fp8_pipe.log"
VLO_LONG_TESTS=1 timeout 900 python -m pytest tests/test_gpu_long.py -m gpu -q -s -x -k "config3 or tensor_parallel" > "$OUT/long_tests.log" 2>&1; echo "exit $?" >> "$OUT/long_tests.log"
timeout 400 python tools/rehearse_config4.py --allreduce p2p > "$OUT/rehearse_cfg4_p2p.json" 2> "$OUT/rehearse_cfg4_p2p.err"
timeout 400 python tools/rehearse_config4.py --allreduce kernel > "$OUT/rehearse_cfg4_kernel.json" 2> "$OUT/rehearse_cfg4_kernel.err"
timeout 900 python bench.py --gpus 1 --model llama-3-70b --weight-dtype fp8 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_70b_fp8_k20.json" 2> "$OUT/bench_70b_fp8.err"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-windows 1024,4096,13312 > "$OUT/bench_8b_cpu_windows.json" 2> "$OUT/bench_8b_cpu_windows.err"
cat "$OUT/fp8_pipe_ab.txt" | grep "gate_up\|total"; tail -3 "$OUT/test_fp8_pipe.log"; grep "^\[" "$OUT/long_tests.log"; tail -3 "$OUT/long_tests.log" | cut -c1-300
cat "$OUT/rehearse_cfg4_p2p.json" "$OUT/rehearse_cfg4_kernel.json"; tail -2 "$OUT/rehearse_cfg4_p2p.err"
python - <<PY
import json
for f in ("bench_70b_fp8_k20", "bench_8b_cpu_windows"):
    try:
        d=json.loads(open("$OUT/"+f+".json").read().strip().splitlines()[-1])
        print(f, d["value"], "fps p50", d["p50_frame_latency_ms"], "p95", d["p95_frame_latency_ms"], "full", d.get("full_stream",{}).get("frames_per_s"), "hbm", d["stream_hbm_roofline"]["frac_of_hbm_peak"], "roof", d["roofline"]["frac"], "enc", d["encode_stage"]["frac_of_mfma_peak"], "cpu", d.get("cpu_baseline"))
    except Exception as ex:
        print(f, "FAILED", ex); print(open("$OUT/"+f.replace("_k20","")+".err").read()[-1200:])
PY
exit 0
