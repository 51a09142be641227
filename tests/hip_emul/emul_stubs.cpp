// The dynamic shared memory arrays the kernels declare with `extern __shared__` (tests/hip_emul/hip_emul.h: one block runs at
// a time, so one process-wide array per name is the block's LDS).
#include <hip/hip_runtime.h>

alignas(16) float4 red[160 * 1024 / 16];        // gemv.hip / prefill.hip
alignas(16) float4 lds_o[160 * 1024 / 16];      // llm_ops.hip (attention)
alignas(16) float4 lds4[160 * 1024 / 16];       // vit.hip (attention)
alignas(16) float prob[160 * 1024 / 4];         // vit.hip (MAP head attention)
alignas(16) uint8_t row[160 * 1024];                // ingest.hip (one input row of the horizontal pass)
