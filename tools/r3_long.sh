#!/usr/bin/env bash
# Round-3 long parity runs (tests/test_gpu_long.py) on the GPU box — ~23 GPU-minutes for the whole file (R3ARGS='-k scheduled' etc. to pick one); the printed numbers go to profiles/r3_parity_measurements.txt
set -u
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/${R3OUT:-r3long}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 VLO_LONG_TESTS=1
timeout 2400 python -m pytest tests/test_gpu_long.py -m gpu -q -s -x ${R3ARGS:-} > "$OUT/long_tests.log" 2>&1; echo "exit $?" >> "$OUT/long_tests.log"
grep "^\[" "$OUT/long_tests.log"; tail -5 "$OUT/long_tests.log" | cut -c1-300
exit 0
