// vit.h — SigLIP vision tower entry points (vit.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "engine.h"

int vit_finalize(vlo_engine *e);
int vit_visual_embed(vlo_engine *e, const uint8_t *frames_dev, int B, void *out_dev, hipStream_t st);
int vit_vision_tokens(vlo_engine *e, const uint8_t *frames_dev, int B, void *out_dev, hipStream_t st);
void vit_destroy(vlo_engine *e);
