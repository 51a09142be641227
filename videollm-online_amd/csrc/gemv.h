// gemv.h — host-visible interface of gemv.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { EPI_PARTIAL_F32 = 0, EPI_BF16 = 1, EPI_BF16_GELU_ERF = 2, EPI_SWIGLU = 3 };

struct GemvArgs {
    const void *Wp;               // packed weights (see gemv.hip)
    const unsigned short *x;      // bf16 [16][ldx]
    float *out_f32;               // EPI_PARTIAL_F32: [ksplit][16][ldo]
    unsigned short *out_bf16;     // other epilogues:  [16][ldo]
    const unsigned short *bias;   // bf16 [N] or null (EPI_BF16 / EPI_BF16_GELU_ERF)
    int K, ldx, ldo;
    int NT;                       // column tiles (N padded to 16)
    int N_valid;                  // real N (multiple of 4)
    int n_rows;                   // valid token rows (<= 16)
    int CT;                       // column tiles per block (0 = auto)
};

struct GemvPlan { int NW, KF, ksplit; };

int gemv_plan(int K, bool allow_ksplit, GemvPlan *p);
hipError_t gemv_launch(GemvArgs a, const GemvPlan &p, int epi, hipStream_t st);
// source tiles [0,NT) of row-major W[N_valid][K] -> packed tiles t*tile_stride + tile_offset of Wp
hipError_t pack_weight_launch(const void *W, void *Wp, int N_valid, int K, int NT, int tile_stride, int tile_offset,
                              hipStream_t st);
