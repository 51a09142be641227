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

// v_mfma_f32_16x16x128_f8f6f4 on OCP e4m3 bytes (cbsz = blgp = 0; zero scale operands select the UNSCALED instruction): a lane's operand is
// 32 bytes = row / column l & 15, 32 of the 128 k's.  Which 32 is the caller's business — a dot product does not care about the order of its
// terms as long as A and B agree — so the operand is given as the two 16-byte halves the callers already hold (D layout = the 16x16x32 forms').
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
VLO_DEV f32x4 mfma_fp8_k128(frag_ab a0, frag_ab a1, frag_ab b0, frag_ab b1, f32x4 c) {
    const i32x8_t A = __builtin_shufflevector(__builtin_bit_cast(i32x4_t, a0), __builtin_bit_cast(i32x4_t, a1), 0, 1, 2, 3, 4, 5, 6, 7);
    const i32x8_t B = __builtin_shufflevector(__builtin_bit_cast(i32x4_t, b0), __builtin_bit_cast(i32x4_t, b1), 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 0, 0, 0, 0, 0, 0);
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

// fp8 e4m3 (OCP, gfx950's native format) -> bf16, exact: every e4m3 value is a bf16 value, so the f32 the converter
// returns is truncated, not rounded.  One 16-byte register = the 8 + 8 weights of two consecutive MFMA fragments.
typedef __attribute__((ext_vector_type(2))) float f32x2_cv;
template <bool HI>
VLO_DEV unsigned fp8x2_to_bf16x2(unsigned src) {          // bytes 0,1 (HI = false) or 2,3 of src
    const f32x2_cv v = __builtin_amdgcn_cvt_pk_f32_fp8((int)src, HI);
    return (__float_as_uint(v[0]) >> 16) | (__float_as_uint(v[1]) & 0xffff0000u);
}
VLO_DEV void fp8x16_to_bf16(frag_ab raw, frag_ab &f0, frag_ab &f1) {
    const uint4 u = __builtin_bit_cast(uint4, raw);
    f0 = __builtin_bit_cast(frag_ab, make_uint4(fp8x2_to_bf16x2<false>(u.x), fp8x2_to_bf16x2<true>(u.x),
                                                  fp8x2_to_bf16x2<false>(u.y), fp8x2_to_bf16x2<true>(u.y)));
    f1 = __builtin_bit_cast(frag_ab, make_uint4(fp8x2_to_bf16x2<false>(u.z), fp8x2_to_bf16x2<true>(u.z),
                                                  fp8x2_to_bf16x2<false>(u.w), fp8x2_to_bf16x2<true>(u.w)));
}
