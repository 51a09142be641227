#!/usr/bin/env bash
# Round-3 GPU call 1: the ping-pong ViT GEMM on hardware (probe + engine tests + encode sweeps + PMC), and the PMC passes the round-2
# verdict asked for on the fp8 GEMV and on the column-packed attention kernel at 15 k tokens.  Results: gpurun_out/r3c1/.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c1
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 tools/_bin/gemm_probe 8 14 16 28 32 > "$OUT/gemm_probe.txt" 2>&1; echo "probe exit $?" >> "$OUT/gemm_probe.txt"
# engine-level correctness of the new kernel (forced onto every shape it accepts) and with the shipped thresholds
VLO_VIT_PP_MIN_ROWS=1 timeout 900 python -m pytest tests/test_gpu_vit.py -m gpu -x -q > "$OUT/test_vit_pp_forced.log" 2>&1; echo "exit $?" >> "$OUT/test_vit_pp_forced.log"
VLO_VIT_PP_MIN_ROWS=1 VLO_VIT_PP_BM=128 timeout 900 python -m pytest tests/test_gpu_vit.py -m gpu -x -q > "$OUT/test_vit_pp128_forced.log" 2>&1; echo "exit $?" >> "$OUT/test_vit_pp128_forced.log"
timeout 900 python -m pytest tests/test_gpu_vit.py -m gpu -x -q > "$OUT/test_vit_default.log" 2>&1; echo "exit $?" >> "$OUT/test_vit_default.log"
# encode sweeps: shipped defaults, one branch only, old kernels
timeout 300 python tools/probe_vit_b.py 8,14,16,28,32,56 10 > "$OUT/vit_sweep_default.txt" 2>&1
VLO_VIT_SPLIT_MIN=999 timeout 300 python tools/probe_vit_b.py 8,14,16,28,32,56 10 > "$OUT/vit_sweep_single_branch.txt" 2>&1
VLO_VIT_PP=0 timeout 300 python tools/probe_vit_b.py 8,14,16,28,32 10 > "$OUT/vit_sweep_old_kernels.txt" 2>&1
VLO_VIT_SPLIT_MIN=999 VLO_VIT_PP_MIN_ROWS=1 VLO_VIT_PP_BM=128 timeout 300 python tools/probe_vit_b.py 7,8,14,16,28 10 > "$OUT/vit_sweep_single_branch_pp128.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
pmc() {   # name, counters..., -- command
    local name=$1; shift
    local ctrs=()
    while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
    shift
    timeout 300 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d "$OUT/pmc_$name" -o pmc -- "$@" > "$OUT/pmc_$name.log" 2>&1
    python $ROOT/tools/pmc_summary.py "$OUT/pmc_$name" > "$OUT/pmc_$name.csv" 2>>"$OUT/pmc_summary.err"
}
export GEMM_PROBE_ITERS=3
for v in pp256 old128; do
    GEMM_PROBE_ONLY=$v pmc gemm_${v}_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $ROOT/tools/_bin/gemm_probe 28
    GEMM_PROBE_ONLY=$v pmc gemm_${v}_sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- $ROOT/tools/_bin/gemm_probe 28
    GEMM_PROBE_ONLY=$v pmc gemm_${v}_fetch FETCH_SIZE -- $ROOT/tools/_bin/gemm_probe 28
    GEMM_PROBE_ONLY=$v pmc gemm_${v}_tcc TCC_HIT_sum TCC_MISS_sum -- $ROOT/tools/_bin/gemm_probe 28
done
unset GEMM_PROBE_ITERS
# fp8 vs bf16 GEMV: where the waves' cycles go (verdict item 4)
for w in fp8 bf16; do
    pmc gemv_${w}_sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -- python $ROOT/tools/bench_gemv.py 8b $w
    pmc gemv_${w}_fetch FETCH_SIZE -- python $ROOT/tools/bench_gemv.py 8b $w
done
# column-packed attention at 15 k cached tokens: bytes fetched per launch (verdict item 5)
pmc attn15k_fetch FETCH_SIZE -- python $ROOT/tools/probe_step.py --lens 15360 --iters 4
pmc attn15k_tcc TCC_HIT_sum TCC_MISS_sum -- python $ROOT/tools/probe_step.py --lens 15360 --iters 4
cd $ROOT
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
cat "$OUT/gemm_probe.txt" | grep -v "cb[1248] " | cut -c1-110
tail -3 "$OUT"/test_vit_*.log | cut -c1-160
cat "$OUT"/vit_sweep_*.txt | grep "B="
exit 0
