#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/${R3OUT:-r3c6}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 tools/_bin/attn_probe 14 16 28 > "$OUT/attn_probe.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
pmc() {
    local name=$1; shift
    local ctrs=()
    while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
    shift
    timeout 300 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d "$OUT/pmc_$name" -o pmc -- "$@" > "$OUT/pmc_$name.log" 2>&1
    python $ROOT/tools/pmc_summary.py "$OUT/pmc_$name" > "$OUT/pmc_$name.csv" 2>>"$OUT/pmc_summary.err"
}
ATTN_PROBE_ONLY=head pmc attn_sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -- $ROOT/tools/_bin/attn_probe 16
ATTN_PROBE_ONLY=head pmc attn_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -- $ROOT/tools/_bin/attn_probe 16
cd $ROOT
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
cat "$OUT/attn_probe.txt"; cat "$OUT/pmc_attn_sq.csv" "$OUT/pmc_attn_mfma.csv" | cut -c1-160
exit 0
