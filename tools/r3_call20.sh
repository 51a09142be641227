#!/usr/bin/env bash
# fp8 image in the 64-token block path: fp8 GPU suite + teacher-forced prefill rate, blocks vs 16-row chunks, fp8 vs bf16
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c20
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -s > "$OUT/pytest_fp8.log" 2>&1; echo "pytest fp8 exit $?"
tail -3 "$OUT/pytest_fp8.log"; grep "block path\|70b-2l" "$OUT/pytest_fp8.log" | cut -c1-200
for wd in fp8 bf16; do
  timeout 300 python tools/probe_prefill.py --tokens 2048 --weight-dtype $wd 2>&1 | grep "tok/s" | tee -a "$OUT/prefill.txt"
done
VLO_BLOCK_PATH=0 timeout 300 python tools/probe_prefill.py --tokens 2048 --weight-dtype fp8 2>&1 | grep "tok/s" | tee -a "$OUT/prefill.txt"
exit 0
