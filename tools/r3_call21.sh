#!/usr/bin/env bash
# SigLIP-so400m/14-384 tower (BASELINE.json configs[4]): parity cases + first timings
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3c21
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_vit.py -x -q -s > "$OUT/pytest_vit.log" 2>&1; echo "pytest vit exit $?"
tail -3 "$OUT/pytest_vit.log"; grep "so400m" "$OUT/pytest_vit.log" | cut -c1-200
VLO_PROBE_VIT=so400m timeout 300 python tools/probe_vit_b.py 1,2,4,8,16,28,56 10 2>&1 | grep "B=\|rror" | tee "$OUT/so400m_sweep.txt"
exit 0
