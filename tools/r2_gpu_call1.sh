#!/usr/bin/env bash
# Round-2 GPU call 1: first hardware run of everything written without GPU time in round 1 (promote-or-delete input).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2c1
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
    tail -n 25 "$OUT/$name.log" | sed 's/^/    | /' >> "$OUT/summary.txt"
}
: > "$OUT/summary.txt"
stage fused_rows_tests 400 env VLO_EXPERIMENTAL=1 python -m pytest tests/test_zz_gpu_fused_rows.py -x -q -m gpu -s
stage persistent_tests 600 env VLO_EXPERIMENTAL=1 python -m pytest tests/test_zz_gpu_persistent.py -q -m gpu -s
stage fused_rows_probe 300 python tools/probe_fused_rows.py --iters 40
stage persistent_probe_prefetch 400 python tools/probe_persistent.py --iters 40
stage p2p_inprocess_tests 400 env VLO_EXPERIMENTAL=1 python -m pytest tests/test_zz_gpu_tp_p2p.py -q -m gpu -s -k "logical_ranks or lonely"
stage p2p_two_process_tests 400 env VLO_EXPERIMENTAL=1 python -m pytest tests/test_zz_gpu_tp_p2p.py -q -m gpu -s -k two_processes
stage p2p_bench_two_ranks 400 env VLO_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 120 --warmup 10 --tp --tp-allreduce p2p \
    --model tinyllama-1.1b --no-cpu-baseline
echo "=== summary" | tee -a "$OUT/summary.txt"
for k in "${!RC[@]}"; do echo "$k: exit ${RC[$k]}"; done | sort | tee -a "$OUT/summary.txt"
exit 0
