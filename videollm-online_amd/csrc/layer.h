// layer.h — host-visible interface of layer.hip (one decoder layer as one persistent launch)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemv.h"
#include "llm_ops.h"

struct LayerArgs {
    // the four projections, filled exactly as for gemv_launch (gemv_prepare has set CT / KC); their virtual grids
    GemvArgs qkv, o, gu, down;
    int qkv_gx, o_gx, gu_gx, down_gx, down_gy;
    // phase 0: residual add of the previous layer's down-proj slices + input RMSNorm
    unsigned short *h;
    const float *prev;            // [prev_ks][16][H] fp32 or null (first layer)
    int prev_ks;
    const unsigned short *ln_in;
    unsigned short *x;
    int H, m;
    float eps;
    // attention (geometry from attention_geometry)
    const unsigned short *q;
    KvGeom kv;
    int layer, nh, G, KS, chunk, nsplit, attn_threads;
    long long pos0;
    float scale;
    float *part_o, *part_ml;
    unsigned short *attn_out;
    // grid barrier: monotonic counter, this launch's barriers wait for bar_base + k * gridDim.x (k = 1 .. layer_barriers_per_launch())
    unsigned *bar_counter, *bar_err;
    unsigned bar_base;
    long long bar_timeout_ticks;
    int barrier_kind;             // 0: flat counter; 1: XCD-hierarchical (layer.hip) — bar_xcd must be ZERO when the launch starts
    unsigned *bar_xcd;            // layer_xcd_words() unsigned words
    int prefetch;                 // 1: next-phase weight fragments are put in flight before each grid barrier (stage P1)
};

int layer_barriers_per_launch(int barrier_kind);
int layer_xcd_words(void);
// instantiations: (fragments per wave of the H-long reductions, of the I-long one, head_dim, query heads per wave)
bool layer_kernel_supports(int kf_h, int kf_i, int head_dim, int hpw);
hipError_t layer_launch(LayerArgs &L, int kf_h, int kf_i, int head_dim, int hpw, int nblocks, size_t lds_bytes, hipStream_t st);
// all layers of a step in ONE launch: `layers_dev` = device array of num_layers LayerArgs (layers_dev[0].bar_base counts); the
// launch makes step_barriers_per_launch(num_layers, kind) arrivals per block on the flat counter
int step_barriers_per_launch(int num_layers, int barrier_kind);
hipError_t step_launch(const LayerArgs *layers_dev, int num_layers, int kf_h, int kf_i, int head_dim, int hpw, int nblocks, size_t lds_bytes,
                       hipStream_t st);
