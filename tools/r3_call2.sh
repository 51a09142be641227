#!/usr/bin/env bash
# Round-3 GPU call 2: persistent ping-pong GEMM with the asynchronous epilogue; GEMV blocks-per-CU sweep.  Results: gpurun_out/r3c2/.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/${R3OUT:-r3c2}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 tools/_bin/gemm_probe 8 14 16 28 32 > "$OUT/gemm_probe.txt" 2>&1; echo "probe exit $?" >> "$OUT/gemm_probe.txt"
VLO_VIT_PP_MIN_ROWS=1 timeout 900 python -m pytest tests/test_gpu_vit.py -m gpu -x -q > "$OUT/test_vit_pp_forced.log" 2>&1; echo "exit $?" >> "$OUT/test_vit_pp_forced.log"
VLO_VIT_PP_MIN_ROWS=1 VLO_VIT_PP_BM=128 timeout 900 python -m pytest tests/test_gpu_vit.py -m gpu -x -q > "$OUT/test_vit_pp128_forced.log" 2>&1; echo "exit $?" >> "$OUT/test_vit_pp128_forced.log"
timeout 900 python -m pytest tests/test_gpu_vit.py -m gpu -x -q > "$OUT/test_vit_default.log" 2>&1; echo "exit $?" >> "$OUT/test_vit_default.log"
timeout 300 python tools/probe_vit_b.py 8,14,16,28,32,56 10 > "$OUT/vit_sweep_default.txt" 2>&1
VLO_VIT_SPLIT_MIN=999 timeout 300 python tools/probe_vit_b.py 8,14,16,28,32,56 10 > "$OUT/vit_sweep_single_branch.txt" 2>&1
VLO_VIT_SPLIT_MIN=999 VLO_VIT_PP_BM=128 timeout 300 python tools/probe_vit_b.py 7,8,14,16,28 10 > "$OUT/vit_sweep_single_branch_pp128.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_vit_b28" -o vit -- python $ROOT/tools/probe_vit_b.py 28 6 > "$OUT/prof_vit_b28.log" 2>&1
cd $ROOT
VLO_VIT_SPLIT_MIN=999 true
db=$(find "$OUT/prof_vit_b28" -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$OUT/kernel_stats_vit_b28.csv"
find "$OUT" -name "*.db" -delete
grep -v "cb[1248] " "$OUT/gemm_probe.txt" | cut -c1-110
for f in "$OUT"/test_vit_*.log; do tail -n 3 "$f" | cut -c1-160; done
for f in "$OUT"/vit_sweep_*.txt; do echo "$f"; grep "B=" "$f"; done
head -25 "$OUT/kernel_stats_vit_b28.csv" | cut -c1-160
exit 0
