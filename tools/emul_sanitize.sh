#!/usr/bin/env bash
# Runs the CPU emulation of the engine's kernels (tests/hip_emul/) under AddressSanitizer + UBSan and under
# ThreadSanitizer: out-of-bounds or misaligned accesses in a kernel's index arithmetic, undefined behaviour, and data
# races between the threads of a block (a missing __syncthreads, two lanes writing one LDS word) show up as reports.
#
#   bash tools/emul_sanitize.sh            # ~25 minutes on 8 cores (two sanitized builds of every kernel source)
#
# Reports that mention only libtorch / libgomp frames (the oracle's OpenMP threads are not instrumented) are noise.
set -u
cd "$(dirname "$0")/.."
RT=$(dirname "$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)")
export VLO_EMUL_FULL=1
mkdir -p gpurun_out
echo "== address,undefined"
VLO_EMUL_SANITIZE=address,undefined LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_emul_llm_path_cpu.py tests/test_tp_p2p_kernels_emul_cpu.py -x -q -s > gpurun_out/emul_asan.log 2>&1
echo "   exit $?; reports:"; grep -E "runtime error|ERROR: AddressSanitizer" gpurun_out/emul_asan.log | sort | uniq -c
echo "== thread"
VLO_EMUL_SANITIZE=thread LD_PRELOAD=$RT/libclang_rt.tsan-x86_64.so TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=2 exitcode=0" \
    python -m pytest tests/test_emul_llm_path_cpu.py tests/test_tp_p2p_kernels_emul_cpu.py -x -q -s -k "not between_processes" > gpurun_out/emul_tsan.log 2>&1
echo "   exit $?; reports outside libtorch:"; grep -E "^SUMMARY" gpurun_out/emul_tsan.log | grep -v "at::native\|hipMemcpy" | sort | uniq -c
