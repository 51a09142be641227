// common.cuh — shared device helpers for the gfx950 (CDNA4 / MI355X) kernels.
// Wave = 64 lanes everywhere in this tree; MFMA fragments follow the gfx950
// v_mfma_f32_16x16x32_{bf16,f16} register maps:
//   A[m][k]: lane l holds m = l&15, k = (l>>4)*8 + j   (j = 0..7, 16 B)
//   B[k][n]: lane l holds n = l&15, k = (l>>4)*8 + j
//   D[m][n]: lane l holds n = l&15, m = (l>>4)*4 + r   (r = 0..3)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;   // raw bfloat16 bits
typedef unsigned short f16_t;    // raw IEEE half bits

typedef __attribute__((ext_vector_type(8))) short  frag_ab;    // 8 x 16-bit MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(8))) __bf16 frag_bf;    // same bits, bf16-typed for the builtin
typedef __attribute__((ext_vector_type(8))) _Float16 frag_h;   // f16-typed
typedef __attribute__((ext_vector_type(4))) float  f32x4;

#define VLO_DEV __device__ __forceinline__

VLO_DEV float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }

// round-to-nearest-even float -> bf16 (what torch's .to(bfloat16) does).  clang lowers these casts to
// gfx950's v_cvt_pk_bf16_f32 and — unlike inline asm — schedules the VALU->MFMA operand hazard itself.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
VLO_DEV unsigned pack2bf(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
VLO_DEV bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
// value after a bf16 rounding point, kept in a float register
VLO_DEV float rbf(float f) { return (float)(__bf16)f; }

VLO_DEV float h2f(f16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
VLO_DEV f16_t f2h(float f) { return __builtin_bit_cast(f16_t, (_Float16)f); }

VLO_DEV f32x4 mfma_bf16(frag_ab a, frag_ab b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(frag_bf, a), __builtin_bit_cast(frag_bf, b), c, 0, 0, 0);
}
VLO_DEV f32x4 mfma_f16(frag_ab a, frag_ab b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(frag_h, a), __builtin_bit_cast(frag_h, b), c, 0, 0, 0);
}

VLO_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
VLO_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blocks of up to 1024 threads; `sm` needs 16 floats
VLO_DEV float block_sum(float v, float *sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += sm[i];
    return t;
}
VLO_DEV float block_max(float v, float *sm) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int i = 0; i < nw; ++i) t = fmaxf(t, sm[i]);
    return t;
}
