// vit.hip — placeholder until the SigLIP kernels land (next commit)
#include "vit.h"
int vit_finalize(vlo_engine *) { return VLO_E_UNSUPPORTED; }
int vit_visual_embed(vlo_engine *, const uint8_t *, int, void *, hipStream_t) { return VLO_E_UNSUPPORTED; }
void vit_destroy(vlo_engine *) {}
