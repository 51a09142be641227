#!/usr/bin/env bash
# Round-3 evidence run (one GPU call): driver bench line, rocprofv3 kernel stats, PMC passes on the ViT GEMMs / attention / gate-up GEMV.
# Summaries land in gpurun_out/r3ev and are copied into profiles/ by hand.  PMC passes are separate runs with --kernel-trace only.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3ev
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_line_k20.json" 2> "$OUT/bench_driver_line.err"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --weight-dtype fp8 --no-cpu-baseline > "$OUT/bench_fp8_weights_k20.json" 2> "$OUT/bench_fp8.err"
timeout 200 tools/_bin/gemm_probe 8 16 28 56 > "$OUT/gemm_probe.txt" 2>&1
timeout 200 tools/_bin/attn_probe 8 16 28 56 > "$OUT/attn_probe.txt" 2>&1
timeout 300 python tools/probe_vit_b.py 1,2,4,8,14,16,28,32,56 10 > "$OUT/vit_batch_sweep.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_bench200" -o b200 -- python $ROOT/bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline > "$OUT/prof_bench200.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_vit_b28" -o vit -- python $ROOT/tools/probe_vit_b.py 28 10 > "$OUT/prof_vit_b28.log" 2>&1
pmc() {
    local name=$1; shift
    local ctrs=()
    while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
    shift
    timeout 300 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d "$OUT/pmc_$name" -o pmc -- "$@" > "$OUT/pmc_$name.log" 2>&1
    python $ROOT/tools/pmc_summary.py "$OUT/pmc_$name" > "$OUT/pmc_$name.csv" 2>>"$OUT/pmc_summary.err"
}
pmc mfma_vit_b28 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python $ROOT/tools/probe_vit_b.py 28 4
pmc sq_vit_b28 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -- python $ROOT/tools/probe_vit_b.py 28 4
pmc fetch_vit_b28 FETCH_SIZE -- python $ROOT/tools/probe_vit_b.py 28 4
pmc tcc_vit_b28 TCC_HIT_sum TCC_MISS_sum -- python $ROOT/tools/probe_vit_b.py 28 4
pmc fetch_llm FETCH_SIZE -- python $ROOT/tools/probe_llm.py --frames 24
pmc write_llm WRITE_SIZE -- python $ROOT/tools/probe_llm.py --frames 24
cd $ROOT
db=$(find "$OUT/prof_bench200" -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$OUT/kernel_stats_bench200.csv"
db=$(find "$OUT/prof_vit_b28" -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$OUT/kernel_stats_vit_b28.csv"
python tools/pmc_hbm_json.py "$OUT/pmc_fetch_llm.csv" "$OUT/pmc_write_llm.csv" "$OUT/pmc_hbm_llm_step.csv" "$OUT/pmc_gemv_gate_up.json" > "$OUT/pmc_hbm.log" 2>&1
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
python - <<PY
import json
for f in ("bench_driver_line_k20", "bench_fp8_weights_k20"):
    try:
        d=json.loads(open("$OUT/"+f+".json").read().strip().splitlines()[-1])
        print(f, d["value"], "fps p50", d["p50_frame_latency_ms"], "p95", d["p95_frame_latency_ms"], "enc", d["encode_stage"]["frac_of_mfma_peak"], "full", d.get("full_stream",{}).get("frames_per_s"), "hbm", d["stream_hbm_roofline"]["frac_of_hbm_peak"], "roof", d["roofline"]["frac"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as ex:
        print(f, "FAILED", ex)
PY
grep "B=" "$OUT/vit_batch_sweep.txt"; cat "$OUT/pmc_hbm.log"; head -10 "$OUT/kernel_stats_bench200.csv" | cut -c1-140; head -9 "$OUT/kernel_stats_vit_b28.csv" | cut -c1-140
grep "vit_gemm_pp\|vit_attn_head" "$OUT/pmc_mfma_vit_b28.csv" "$OUT/pmc_sq_vit_b28.csv" "$OUT/pmc_fetch_vit_b28.csv" "$OUT/pmc_tcc_vit_b28.csv" | cut -c1-200
exit 0
