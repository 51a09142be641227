#!/usr/bin/env bash
# First GPU call of the next round: validate and time everything that was written without GPU time.
#
#   gpurun --timeout 2400 -- 'bash tools/next_round_gpu_checks.sh'
#
# Every stage runs under its own `timeout`, writes its log to gpurun_out/next_round/ and never stops the later
# stages; the summary at the end says which stage passed.  Nothing here changes a default: the fused short-chunk
# pipeline (VLO_FUSED_ROWS) and the peer-to-peer tensor-parallel exchange (allreduce="p2p") stay opt-in until these
# logs say they are correct AND faster.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/next_round
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
declare -A RC

stage() {   # name, seconds, command...
    local name=$1 secs=$2; shift 2
    echo "=== $name" | tee -a "$OUT/summary.txt"
    timeout "$secs" "$@" > "$OUT/$name.log" 2>&1
    RC[$name]=$?
    echo "    exit ${RC[$name]}" | tee -a "$OUT/summary.txt"
    tail -n 15 "$OUT/$name.log" | sed 's/^/    | /' >> "$OUT/summary.txt"
}

: > "$OUT/summary.txt"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1 || { echo "build failed" | tee -a "$OUT/summary.txt"; exit 1; }

# 1. the default suite's TP tests first: tp_chunk was refactored around tp_reduce_norm (same launches, same order)
stage default_tp_tests 600 python -m pytest tests/test_gpu_tp.py -x -q -m gpu
# 2. fused short-chunk pipeline: parity vs oracle and vs the default pipeline, then the A/B timing
stage fused_rows_tests 600 env VLO_EXPERIMENTAL=1 python -m pytest tests/test_zz_gpu_fused_rows.py -x -q -m gpu -s
stage fused_rows_probe 420 python tools/probe_fused_rows.py --iters 40
# 3. peer-to-peer exchange: logical ranks in one process, then two processes sharing the GPU over hipIpc
stage p2p_inprocess_tests 600 env VLO_EXPERIMENTAL=1 python -m pytest tests/test_zz_gpu_tp_p2p.py -x -q -m gpu -s -k "logical_ranks or lonely"
stage p2p_two_process_tests 600 env VLO_EXPERIMENTAL=1 python -m pytest tests/test_zz_gpu_tp_p2p.py -x -q -m gpu -s -k two_processes
# 4. the whole multi-process TP bench path on ONE GPU (gloo control plane, two ranks share GPU 0): p2p needs no RCCL, so
#    this runs end to end here; the RCCL variant is expected to refuse the duplicate GPU
stage p2p_bench_two_ranks 600 env VLO_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 120 --warmup 10 --tp --tp-allreduce p2p \
    --model tinyllama-1.1b --no-cpu-baseline
# 4b. persistent layer kernel: bit-identical to the launch pipeline?  then A/B timing with and without the cross-phase prefetch
stage persistent_tests 900 env VLO_EXPERIMENTAL=1 python -m pytest tests/test_zz_gpu_persistent.py -x -q -m gpu -s
stage persistent_probe_prefetch 420 python tools/probe_persistent.py --iters 40
stage persistent_probe_noprefetch 420 env VLO_PERSISTENT_PREFETCH=0 python tools/probe_persistent.py --iters 40
# 5. end-to-end effect of the fused decode path on the headline bench (short run), default vs VLO_FUSED_ROWS=1
stage bench_default_300 420 python bench.py --steps 300 --warmup 10 --no-cpu-baseline
stage bench_fused1_300 420 env VLO_FUSED_ROWS=1 python bench.py --steps 300 --warmup 10 --no-cpu-baseline
stage bench_persistent_300 420 env VLO_PERSISTENT=1 python bench.py --steps 300 --warmup 10 --no-cpu-baseline
stage bench_persistent_step_300 420 env VLO_PERSISTENT=1 VLO_PERSISTENT_STEP=1 python bench.py --steps 300 --warmup 10 --no-cpu-baseline
stage bench_persistent_step_xcd_300 420 env VLO_PERSISTENT=1 VLO_PERSISTENT_STEP=1 VLO_PERSISTENT_BARRIER=xcd python bench.py --steps 300 --warmup 10 --no-cpu-baseline

echo "=== summary" | tee -a "$OUT/summary.txt"
for k in "${!RC[@]}"; do echo "$k: exit ${RC[$k]}"; done | sort | tee -a "$OUT/summary.txt"
grep -h '^{' "$OUT"/bench_*.log "$OUT"/p2p_bench_two_ranks.log 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    try:
        d = json.loads(l)
        print('bench:', d['config']['parallelism'], d['value'], d['unit'], 'p50', d.get('p50_frame_latency_ms'), 'ms')
    except Exception as ex:
        print('unparsed bench line', ex)
" | tee -a "$OUT/summary.txt"
exit 0
