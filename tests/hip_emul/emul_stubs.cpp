// Pieces the emulated library needs besides the engine's own sources (tests/hip_emul/hip_emul.h):
//   * the dynamic shared memory arrays the kernels declare with `extern __shared__` (one block runs at a time);
//   * stubs for the vision tower (csrc/vit.hip is not emulated: fp16 MFMA GEMMs with direct-to-LDS loads and a hipGraph).
#include <hip/hip_runtime.h>

#include "../../include/vlo.h"
#include "engine.h"
#include "vit.h"

alignas(16) float4 red[160 * 1024 / 16];        // gemv.hip / prefill.hip
alignas(16) float4 lds_o[160 * 1024 / 16];      // llm_ops.hip (attention)

static int no_vit() { return vlo_fail(VLO_E_UNSUPPORTED, "the vision tower is not part of the CPU emulation"); }
int vit_finalize(vlo_engine *) { return no_vit(); }
void vit_destroy(vlo_engine *) {}
int vit_visual_embed(vlo_engine *, const uint8_t *, int, void *, hipStream_t) { return no_vit(); }
int vit_vision_tokens(vlo_engine *, const uint8_t *, int, void *, hipStream_t) { return no_vit(); }
