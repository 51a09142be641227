// attn_probe.hip — the SigLIP-L self-attention kernels (csrc/vit_attn.inc) outside the engine: time per launch, TFLOP/s (4 S^2 hd per
// head and frame), the whole-head kernel against the 64-query-tile kernel (max abs difference), and two ablations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/attn_probe.hip -o tools/_bin/attn_probe;   attn_probe [frames ...]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../videollm-online_amd/csrc/common.cuh"
VLO_DEV int vt_pos(int t) { return (t & ~31) | (((t >> 2) & 3) << 3) | (((t >> 4) & 1) << 2) | (t & 3); }
#include "../videollm-online_amd/csrc/vit_attn.inc"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static unsigned long long rs = 88172645463325252ull;
static float urand() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (float)((rs >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; }
static f16_t h16(float f) { _Float16 t = (_Float16)f; f16_t r; memcpy(&r, &t, 2); return r; }
static float f16f(f16_t h) { _Float16 t; memcpy(&t, &h, 2); return (float)t; }

int main(int argc, char **argv) {
    std::vector<int> frames;
    for (int i = 1; i < argc; ++i) frames.push_back(atoi(argv[i]));
    if (frames.empty()) frames = {8, 14, 16, 28, 32};
    const int S = 576, D = 1024, NH = 16, HD = 64, Sp = 576;
    int maxB = 0;
    for (int b : frames) maxB = b > maxB ? b : maxB;
    const size_t M = (size_t)maxB * S;
    std::vector<f16_t> hqk(M * 2 * D), hvt((size_t)maxB * D * Sp);
    for (auto &x : hqk) x = h16(urand() * 2.0f);
    for (auto &x : hvt) x = h16(urand());
    f16_t *qk, *vt, *o0, *o1;
    CK(hipMalloc(&qk, hqk.size() * 2)); CK(hipMalloc(&vt, hvt.size() * 2)); CK(hipMalloc(&o0, M * D * 2)); CK(hipMalloc(&o1, M * D * 2));
    CK(hipMemcpy(qk, hqk.data(), hqk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(vt, hvt.data(), hvt.size() * 2, hipMemcpyHostToDevice));
    const size_t lds_old = (size_t)4 * 4 * 4 * 64 * 16 + 4 * 4 * 16 * 2 * 4;
    const int cpr = Sp / 8, vrs = (cpr + ((10 - cpr % 16) + 16) % 16) * 16;
    const size_t lds_new = (size_t)Sp * 128 + (size_t)64 * vrs;
    CK(hipFuncSetAttribute((const void *)vit_attn_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_old));
    CK(hipFuncSetAttribute((const void *)vit_attn_head_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_new));
    CK(hipFuncSetAttribute((const void *)vit_attn_head_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_new));
    CK(hipFuncSetAttribute((const void *)vit_attn_head_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_new));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const float scale = 0.125f;
    for (int B : frames) {
        auto run = [&](int which, f16_t *out) {
            if (which == 0) hipLaunchKernelGGL((vit_attn_kernel<64>), dim3((S + 63) / 64, NH, B), dim3(256), lds_old, st, qk, vt, out, S, D, NH, scale);
            else if (which == 1) hipLaunchKernelGGL((vit_attn_head_kernel<0>), dim3(1, NH, B), dim3(768), lds_new, st, qk, vt, out, S, D, NH, scale * 1.4426950408889634f, vrs);
            else if (which == 2) hipLaunchKernelGGL((vit_attn_head_kernel<1>), dim3(1, NH, B), dim3(768), lds_new, st, qk, vt, out, S, D, NH, scale * 1.4426950408889634f, vrs);
            else hipLaunchKernelGGL((vit_attn_head_kernel<2>), dim3(1, NH, B), dim3(768), lds_new, st, qk, vt, out, S, D, NH, scale * 1.4426950408889634f, vrs);
        };
        const char *names[4] = {"tile64", "head", "head-nosoftmax", "head-nofill"};
        std::vector<f16_t> r0((size_t)B * S * D), r1((size_t)B * S * D);
        for (int which = 0; which < 4; ++which) {
            if (getenv("ATTN_PROBE_ONLY") && strcmp(getenv("ATTN_PROBE_ONLY"), names[which])) continue;
            f16_t *out = which == 0 ? o0 : o1;
            run(which, out);
            CK(hipStreamSynchronize(st));
            CK(hipGetLastError());
            double maxd = -1;
            if (which <= 1) {
                CK(hipMemcpy((which == 0 ? r0 : r1).data(), out, r0.size() * 2, hipMemcpyDeviceToHost));
                if (which == 1) {
                    maxd = 0;
                    for (size_t i = 0; i < r0.size(); ++i) maxd = fmax(maxd, fabs((double)f16f(r0[i]) - (double)f16f(r1[i])));
                }
            }
            for (int i = 0; i < 3; ++i) run(which, out);
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 20; ++i) run(which, out);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / 20, flop = 4.0 * S * S * HD * NH * B;
            printf("B=%2d %-15s %8.1f us %7.0f TFLOP/s", B, names[which], us, flop / us / 1e6);
            if (maxd >= 0) printf("   max |head - tile64| = %.4g", maxd);
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}
