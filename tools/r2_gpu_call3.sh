#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2c3
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 120 tools/chain_probe 32 5 > "$OUT/chain_probe.log" 2>&1; echo "chain_probe exit $?" >> "$OUT/chain_probe.log"
timeout 1500 python -m pytest tests -m gpu -q -s > "$OUT/gpu_suite.log" 2>&1; echo "gpu_suite exit $?" >> "$OUT/gpu_suite.log"
tail -5 "$OUT/chain_probe.log"; tail -30 "$OUT/gpu_suite.log"
exit 0
