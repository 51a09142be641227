#!/usr/bin/env bash
# Round-2 evidence run (one GPU call): bench lines, rocprofv3 kernel stats, PMC passes; summaries land in gpurun_out/r2ev and are
# copied into profiles/ by hand.  PMC passes are separate runs with --kernel-trace only (gpurun refuses other trace domains).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2ev
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_line.json" 2> "$OUT/bench_driver_line.err"
timeout 400 python bench.py --gpus 1 --steps 1200 --warmup 5 --no-cpu-baseline > "$OUT/bench_full_stream_1200.json" 2> "$OUT/bench_full.err"
timeout 300 python bench.py --gpus 1 --model tinyllama-1.1b --steps 60 --warmup 5 --no-cpu-baseline > "$OUT/bench_cfg1_tinyllama_60frames.json" 2> "$OUT/bench_cfg1.err"
timeout 500 python bench.py --gpus 1 --fps 10 --steps 200 --warmup 5 --no-cpu-baseline > "$OUT/bench_cfg3_10fps_last200_of_6000.json" 2> "$OUT/bench_cfg3.err"
timeout 200 python tools/probe_vit_b.py 1,4,8,12,16,32 10 > "$OUT/vit_batch_sweep.txt" 2>&1
timeout 200 python tools/probe_step.py --lens 0,4096,12288,15360,61440 > "$OUT/llm_step_bf16.txt" 2>&1
timeout 200 python tools/probe_step.py --weight-dtype fp8 --lens 0,4096,12288,15360,61440 > "$OUT/llm_step_fp8.txt" 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --weight-dtype fp8 --no-cpu-baseline > "$OUT/bench_fp8_k20.json" 2> "$OUT/bench_fp8.err"
timeout 200 python tools/bench_gemv.py 8b both > "$OUT/bench_gemv_8b.txt" 2>&1
timeout 200 python tools/bench_gemv.py 70b-tp8 both > "$OUT/bench_gemv_70b_tp8.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_bench200" -o b200 -- python $ROOT/bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline > "$OUT/prof_bench200.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python $ROOT/tools/probe_llm.py --frames 24 > "$OUT/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python $ROOT/tools/probe_llm.py --frames 24 > "$OUT/pmc_write.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -o pmc -- python $ROOT/tools/probe_vit_b.py 16 4 > "$OUT/pmc_mfma.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_vit_b16" -o vit -- python $ROOT/tools/probe_vit_b.py 16 10 > "$OUT/prof_vit_b16.log" 2>&1
cd $ROOT
db=$(find "$OUT/prof_bench200" -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$OUT/kernel_stats_bench200.csv"
db=$(find "$OUT/prof_vit_b16" -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$OUT/kernel_stats_vit_b16.csv"
for d in pmc_fetch pmc_write pmc_mfma; do python tools/pmc_summary.py "$OUT/$d" > "$OUT/$d.csv" 2>>"$OUT/pmc_summary.err"; done
python tools/pmc_hbm_json.py "$OUT/pmc_fetch.csv" "$OUT/pmc_write.csv" "$OUT/pmc_hbm_llm_step.csv" "$OUT/pmc_gemv_gate_up.json" > "$OUT/pmc_hbm.log" 2>&1
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/gpu_suite.log" 2>&1; echo "gpu_suite exit $?" >> "$OUT/gpu_suite.log"
for f in bench_driver_line bench_full_stream_1200 bench_cfg1_tinyllama_60frames bench_cfg3_10fps_last200_of_6000; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    print("$f", d["value"], "fps p50", d["p50_frame_latency_ms"], "p95", d["p95_frame_latency_ms"], "enc", d["encode_stage"]["frac_of_mfma_peak"], "full", d.get("full_stream",{}).get("frames_per_s"), "hbm", d["stream_hbm_roofline"]["frac_of_hbm_peak"], "roof", d["roofline"]["frac"], "cpu", (d.get("cpu_baseline") or {}).get("value"), "kv", d["config"]["final_kv_tokens"])
except Exception as ex:
    print("$f", "FAILED", ex)
PY
done
cat "$OUT/vit_batch_sweep.txt" | grep "B="; cat "$OUT/pmc_hbm.log"; head -12 "$OUT/kernel_stats_bench200.csv" | cut -c1-140; head -10 "$OUT/pmc_mfma.csv" | cut -c1-150; tail -4 "$OUT/gpu_suite.log" | cut -c1-200
exit 0
