// mfma_probe.hip — the matrix pipe's rate on THIS box for the two fp16 / bf16 MFMA shapes, register-resident (no memory, no LDS): round-5 verdict item 4
// ("every MFMA in the repo is 16x16x32: the guide's table gives ~5 cycles per CU against 8 for 32x32x16 at twice the FLOPs — measure before arguing").
// Each wave runs NACC independent accumulator chains of one shape for `iters` rounds; 256 workgroups x {4, 8} waves (1 or 2 waves per SIMD).
// Output: TFLOP/s chip-wide and cycles per MFMA per SIMD at a nominal 2.4 GHz (the chip's real clock under this load is lower: compare the two shapes
// with each other, not with the data sheet).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_probe.hip -o tools/_bin/mfma_probe ; tools/_bin/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(8))) __bf16 b8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(8))) int i8v;

// SHAPE 0: v_mfma_f32_16x16x32_f16 (16 K FLOP), 1: v_mfma_f32_32x32x16_f16 (32 K FLOP), 2 / 3: the bf16 forms,
// 4: v_mfma_f32_16x16x128_f8f6f4 on e4m3 operands (64 K FLOP; BASELINE.json configs[4]'s "fp8 MFMA"), 5: v_mfma_f32_16x16x32_fp8_fp8 (16 K FLOP)
template <int SHAPE, int NACC, int NT>
__global__ __launch_bounds__(NT) void mfma_kernel(float *out, int iters, float seed) {
    h8 a, b;
    b8 ab, bb;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(seed * (float)((threadIdx.x + i) & 7) * 0.01f);
        b[i] = (_Float16)(seed * (float)((threadIdx.x * 3 + i) & 7) * 0.01f);
        ab[i] = (__bf16)(float)a[i];
        bb[i] = (__bf16)(float)b[i];
    }
    float r = 0.f;
    if (SHAPE == 6) {            // v_mfma_f32_32x32x64_f8f6f4 on e4m3 operands (128 K FLOP)
        i8v qa, qb;
#pragma unroll
        for (int i = 0; i < 8; ++i) { qa[i] = 0x38383838 + (int)threadIdx.x * 0x01010101 * (i & 1); qb[i] = 0x3c343c34 ^ ((int)threadIdx.x << (i & 3)); }
        f16v acc[NACC];
#pragma unroll
        for (int j = 0; j < NACC; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc[j], 0, 0, 0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NACC; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) r += acc[j][e];
    } else if (SHAPE == 4 || SHAPE == 5) {
        i8v qa, qb;
#pragma unroll
        for (int i = 0; i < 8; ++i) { qa[i] = 0x38383838 + (int)threadIdx.x * 0x01010101 * (i & 1); qb[i] = 0x3c343c34 ^ ((int)threadIdx.x << (i & 3)); }
        f4 acc[NACC];
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};
        const long qa8 = ((long)qa[1] << 32) | (unsigned)qa[0], qb8 = ((long)qb[1] << 32) | (unsigned)qb[0];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                if (SHAPE == 4) acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(qa, qb, acc[j], 0, 0, 0, 0, 0, 0);
                else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(qa8, qb8, acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NACC; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else if (SHAPE == 0 || SHAPE == 2) {
        f4 acc[NACC];
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                if (SHAPE == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
                else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NACC; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else {
        f16v acc[NACC];
#pragma unroll
        for (int j = 0; j < NACC; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                if (SHAPE == 1) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
                else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NACC; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) r += acc[j][e];
    }
    if (r == 1.2345f) out[0] = r;
}

template <int SHAPE, int NACC, int NT>
static void run(const char *name, float *out, hipStream_t st) {
    const int iters = 20000 / NACC * 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((mfma_kernel<SHAPE, NACC, NT>), dim3(256), dim3(NT), 0, st, out, iters, 1.0f);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        if (rep == 1) {
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double n_mfma = 256.0 * (NT / 64) * (double)iters * NACC;
            const double flop = n_mfma * (SHAPE == 6 ? 131072.0 : SHAPE == 4 ? 65536.0 : SHAPE == 5 ? 16384.0 : (SHAPE & 1) ? 32768.0 : 16384.0);
            const double per_simd = n_mfma / (256.0 * 4);
            printf("%-28s %d waves/CU, %d chains/wave: %8.3f ms  %7.0f TFLOP/s  %5.2f cycles per MFMA per SIMD @ 2.4 GHz\n", name, NT / 64, NACC, ms, flop / (ms * 1e-3) / 1e12,
                   ms * 1e-3 * 2.4e9 / per_simd);
        }
    }
}

int main() {
    float *out;
    CK(hipMalloc(&out, 64));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    run<0, 4, 256>("v_mfma_f32_16x16x32_f16", out, st);
    run<0, 8, 256>("v_mfma_f32_16x16x32_f16", out, st);
    run<0, 4, 512>("v_mfma_f32_16x16x32_f16", out, st);
    run<0, 8, 512>("v_mfma_f32_16x16x32_f16", out, st);
    run<1, 2, 256>("v_mfma_f32_32x32x16_f16", out, st);
    run<1, 4, 256>("v_mfma_f32_32x32x16_f16", out, st);
    run<1, 2, 512>("v_mfma_f32_32x32x16_f16", out, st);
    run<1, 4, 512>("v_mfma_f32_32x32x16_f16", out, st);
    run<2, 8, 512>("v_mfma_f32_16x16x32_bf16", out, st);
    run<3, 4, 512>("v_mfma_f32_32x32x16_bf16", out, st);
    run<4, 4, 256>("v_mfma_f32_16x16x128_f8f6f4 (e4m3)", out, st);
    run<4, 8, 512>("v_mfma_f32_16x16x128_f8f6f4 (e4m3)", out, st);
    run<5, 8, 512>("v_mfma_f32_16x16x32_fp8_fp8", out, st);
    // ONE wave per SIMD (what a ping-pong schedule's MFMA segment sees: the partner wave is reading), every shape
    run<2, 4, 256>("v_mfma_f32_16x16x32_bf16", out, st);
    run<2, 8, 256>("v_mfma_f32_16x16x32_bf16", out, st);
    run<3, 2, 256>("v_mfma_f32_32x32x16_bf16", out, st);
    run<3, 4, 256>("v_mfma_f32_32x32x16_bf16", out, st);
    run<6, 2, 256>("v_mfma_f32_32x32x64_f8f6f4 (e4m3)", out, st);
    run<6, 4, 256>("v_mfma_f32_32x32x64_f8f6f4 (e4m3)", out, st);
    run<6, 4, 512>("v_mfma_f32_32x32x64_f8f6f4 (e4m3)", out, st);
    return 0;
}
