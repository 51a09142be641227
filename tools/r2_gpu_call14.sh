#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2c14
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python -m pytest tests/test_gpu_vit.py tests/test_gpu_liveinfer.py tests/test_gpu_ingest.py -m gpu -q -s > "$OUT/vit_tests.log" 2>&1; echo "exit $?" >> "$OUT/vit_tests.log"
timeout 300 python tools/probe_vit.py > "$OUT/probe_vit.log" 2>&1
VLO_VIT_SPLIT_MIN=0 timeout 300 python tools/probe_vit.py > "$OUT/probe_vit_nosplit.log" 2>&1
grep "B=" "$OUT/probe_vit.log"; echo nosplit; grep "B=" "$OUT/probe_vit_nosplit.log"; grep -a "two-branch" "$OUT/vit_tests.log" | head -3; tail -3 "$OUT/vit_tests.log"
exit 0
