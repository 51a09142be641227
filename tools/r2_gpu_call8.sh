#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2c8
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python tools/probe_vit_b.py 1,4,8,14 10 > "$OUT/vit_sweep_w8.log" 2>&1
VLO_VIT_WAVES=4 timeout 200 python tools/probe_vit_b.py 1,4,8,14 10 > "$OUT/vit_sweep_w4.log" 2>&1
VLO_VIT_WAVES=4 VLO_VIT_STAGES=3 timeout 200 python tools/probe_vit_b.py 8,14 10 > "$OUT/vit_sweep_w4s3.log" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_sq" -o pmc -- python $ROOT/tools/probe_vit_b.py 8 3 > "$OUT/pmc_sq.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_tcc" -o pmc -- python $ROOT/tools/probe_vit_b.py 8 3 > "$OUT/pmc_tcc.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python $ROOT/tools/probe_vit_b.py 8 3 > "$OUT/pmc_fetch.log" 2>&1
VLO_VIT_WAVES=4 timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_w4" -o vit -- python $ROOT/tools/probe_vit_b.py 8 10 > "$OUT/prof_w4.log" 2>&1
cd $ROOT
for d in pmc_sq pmc_tcc pmc_fetch; do python tools/pmc_summary.py "$OUT/$d" > "$OUT/$d.csv" 2>>"$OUT/pmc_summary.err"; done
db=$(find "$OUT/prof_w4" -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$OUT/vit_b8_w4_kernel_stats.csv"
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*counter_collection.csv" -size +3M -delete; find "$OUT" -name "*kernel_trace.csv" -delete
echo w8; grep "B=" "$OUT/vit_sweep_w8.log"; echo w4; grep "B=" "$OUT/vit_sweep_w4.log"; echo w4s3; grep "B=" "$OUT/vit_sweep_w4s3.log"
grep "vit_gemm\|vit_attn" "$OUT/pmc_sq.csv" | cut -c1-200; grep "vit_gemm\|vit_attn" "$OUT/pmc_tcc.csv" "$OUT/pmc_fetch.csv" | cut -c1-220
head -7 "$OUT/vit_b8_w4_kernel_stats.csv" | cut -c1-150
exit 0
