#!/usr/bin/env bash
# the GPU suite files that have not run on the final code yet (vit, fp8 and the long file ran separately) + smoke
set -u
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r3c24
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 420 python -m pytest tests/test_gpu_llm.py tests/test_gpu_eval.py tests/test_gpu_liveinfer.py tests/test_gpu_ingest.py tests/test_gpu_tp.py tests/test_gpu_tp_p2p.py -m gpu -x -q > "$OUT/pytest_rest.log" 2>&1; echo "pytest exit $?"
tail -4 "$OUT/pytest_rest.log" | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
exit 0
