#!/usr/bin/env bash
# ONE parametrised evidence script for the GPU box (replaces the per-session tools/r2_*.sh / r3_call*.sh of earlier rounds).
#
#   tools/gpu_evidence.sh OUT 'STEP ...' ['STEP ...' ...]           results land in gpurun_out/OUT/ (merged back by gpurun)
#
# Every STEP is one quoted string, first word = verb:
#   gate                         the driver's gate, literally: `python -m pytest tests -m gpu -x -q`, then __graft_entry__.smoke()
#   pytest NAME ARGS...          python -m pytest ARGS (log: pytest_NAME.log); env assignments may precede via `env`
#   bench NAME ARGS...           python bench.py ARGS > bench_NAME.json (+ a one-line digest on stdout)
#   run NAME CMD...              any command, stdout+stderr -> NAME.txt (probes, sweeps)
#   prof NAME CMD...             rocprofv3 --kernel-trace --stats -- CMD  -> kernel_stats_NAME.csv (tools/rocpd_stats.py)
#   pmc NAME CTR [CTR...] -- CMD one rocprofv3 --kernel-trace --pmc pass (never combined with other trace domains) -> pmc_NAME.csv
#   long [PYTEST ARGS]           the long parity file alone (default-on since round 6; with VLO_FOLLOWER_DEVICE=cpu ~23 GPU-minutes)
# A step's own time limit: prefix the string with `T=<seconds>` (default 600).  Steps never abort the script; each prints its exit code.
#
# Example (one gpurun call):
#   gpurun --timeout 1500 -- "tools/gpu_evidence.sh r4a 'gate' 'bench k20 --gpus 1 --steps 20 --warmup 5' \
#       'prof bench200 python bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline' \
#       'pmc fetch_llm FETCH_SIZE -- python tools/probe_llm.py --frames 24'"
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
[ $# -ge 2 ] || { sed -n 2,22p "$0"; exit 2; }
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp

abspaths() {  # rocprofv3 runs from /tmp: words that name files of the repo become absolute paths
    ABS=(); local w; for w in "$@"; do if [ -e "$ROOT/$w" ] && [[ "$w" != /* ]]; then ABS+=("$ROOT/$w"); else ABS+=("$w"); fi; done
}

digest() {   # one line per bench JSON
    python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    g = lambda *k: (lambda v: v)(__import__("functools").reduce(lambda a, b: (a or {}).get(b), k, d))
    print("   ", d["value"], d["unit"], "| p50", g("p50_frame_latency_ms"), "p95", g("p95_frame_latency_ms"), "| full_stream", g("full_stream", "frames_per_s"),
          "hbm", g("full_stream", "frac_of_hbm_peak"), "| roofline", g("roofline", "frac"), "| encode", g("encode_stage", "frac_of_mfma_peak"),
          "| live_feed", g("live_feed", "frames_per_s"), "| cpu", g("cpu_baseline", "value"))
except Exception as ex:
    print("    bench line unreadable:", ex)
PY
}

for step in "$@"; do
    T=600
    # shellcheck disable=SC2086
    set -- $step
    case "$1" in T=*) T=${1#T=}; shift;; esac
    verb=$1; shift
    echo "== $verb $* (limit ${T}s)"
    case "$verb" in
    gate)
        timeout "$T" python -m pytest tests -m gpu -x -q > "$OUT/pytest_gate.log" 2>&1; echo "   pytest exit $?"
        tail -4 "$OUT/pytest_gate.log" | cut -c1-240
        timeout 180 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "   smoke exit $?"; tail -2 "$OUT/smoke.log" | cut -c1-300 ;;
    pytest)
        name=$1; shift
        timeout "$T" python -m pytest "$@" > "$OUT/pytest_$name.log" 2>&1; echo "   exit $?"
        grep "^\[" "$OUT/pytest_$name.log" | cut -c1-400; tail -3 "$OUT/pytest_$name.log" | cut -c1-240 ;;
    bench)
        name=$1; shift
        timeout "$T" python bench.py "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; echo "   exit $?"
        digest "$OUT/bench_$name.json" ;;
    run)
        name=$1; shift
        timeout "$T" "$@" > "$OUT/$name.txt" 2>&1; echo "   exit $?"; grep -v amdgpu.ids "$OUT/$name.txt" | tail -40 | cut -c1-200 ;;
    prof)
        name=$1; shift
        abspaths "$@"
        (cd /tmp && timeout "$T" rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o p -- "${ABS[@]}" > "$OUT/prof_$name.log" 2>&1); echo "   exit $?"
        db=$(find "$OUT/prof_$name" -name "*.db" | head -1)
        [ -n "$db" ] && (cd "$ROOT" && python tools/rocpd_stats.py "$db" > "$OUT/kernel_stats_$name.csv") && head -14 "$OUT/kernel_stats_$name.csv" | cut -c1-170
        rm -rf "$OUT/prof_$name" ;;
    pmc)
        name=$1; shift
        ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
        abspaths "$@"
        (cd /tmp && timeout "$T" rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d "$OUT/pmcdir_$name" -o pmc -- "${ABS[@]}" > "$OUT/pmc_$name.log" 2>&1); echo "   exit $?"
        (cd "$ROOT" && python tools/pmc_summary.py "$OUT/pmcdir_$name" > "$OUT/pmc_$name.csv" 2>> "$OUT/pmc_summary.err") && head -12 "$OUT/pmc_$name.csv" | cut -c1-200
        rm -rf "$OUT/pmcdir_$name" ;;
    long)
        VLO_LONG_TESTS=1 timeout "$T" python -m pytest tests/test_gpu_long.py -m gpu -q -s -x "$@" > "$OUT/long_tests.log" 2>&1; echo "   exit $?"
        grep "^\[" "$OUT/long_tests.log" | cut -c1-400; tail -3 "$OUT/long_tests.log" | cut -c1-240 ;;
    *) echo "   unknown step verb '$verb'";;
    esac
done
exit 0
