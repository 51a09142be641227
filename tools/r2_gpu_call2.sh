#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2c2
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
declare -A RC
stage() {
    local name=$1 secs=$2; shift 2
    echo "=== $name" | tee -a "$OUT/summary.txt"
    local t0=$SECONDS
    timeout "$secs" "$@" > "$OUT/$name.log" 2>&1
    RC[$name]=$?
    echo "    exit ${RC[$name]} in $((SECONDS-t0))s" | tee -a "$OUT/summary.txt"
    tail -n 30 "$OUT/$name.log" | sed 's/^/    | /' >> "$OUT/summary.txt"
}
: > "$OUT/summary.txt"
stage persistent_tests 400 env VLO_EXPERIMENTAL=1 python -m pytest tests/test_zz_gpu_persistent.py -q -m gpu
stage persistent_probe 400 python tools/probe_persistent.py --iters 40
stage bench_driver_line 600 python bench.py --gpus 1 --steps 20 --warmup 5
stage gpu_suite 1200 python -m pytest tests -m gpu -q -s -x
echo "=== summary" | tee -a "$OUT/summary.txt"
for k in "${!RC[@]}"; do echo "$k: exit ${RC[$k]}"; done | sort | tee -a "$OUT/summary.txt"
exit 0
