// gemm_probe.hip — the SigLIP-L GEMM shapes through the engine's own GEMM kernels (csrc/vit_gemm.inc), variant by variant,
// outside the engine: time per launch (HIP events, 20 launches), TFLOP/s, and a bit-exact check of every variant against the
// 128 x 128 kernel (same MFMA, same k order).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_probe.hip -o tools/_bin/gemm_probe
//   gemm_probe [frames ...]            (default 8 14 16 28 32)
//   GEMM_PROBE_ONLY=name               run one variant only (for rocprofv3 --pmc passes): old128 old256 pp256 pp256np pp128 pp256cbN
//   GEMM_PROBE_GEMM=qkv|out|fc1|fc2    one GEMM only
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../videollm-online_amd/csrc/common.cuh"
#include "../videollm-online_amd/csrc/vit_gemm.inc"

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float urand() {      // uniform [-1, 1)
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (float)((rng_state >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f;
}
static f16_t h16(float f) { _Float16 t = (_Float16)f; f16_t r; memcpy(&r, &t, 2); return r; }

static int g_grid = 256;       // workgroups of the persistent ping-pong kernel (GEMM_PROBE_GRID)
struct Variant { const char *name; int kind, bm, cb, prio; };     // kind 0 = old128, 1 = old256 (16 waves), 2 = ping-pong; prio 2 / 3 = ablations (no epilogue / no loads)

template <int EP>
static void launch_pp(const Variant &v, const GemmArgs &a, hipStream_t st) {
    const int bm = v.bm, nt = std::min(((a.M + bm - 1) / bm) * (a.N / 256), g_grid);
    if (bm == 128) hipLaunchKernelGGL((vit_gemm_pp_kernel<128, EP, 1>), dim3(nt), dim3(512), 0, st, a);
    else if (v.prio == 2) hipLaunchKernelGGL((vit_gemm_pp_kernel<256, EP, 1, 1>), dim3(nt), dim3(512), 0, st, a);
    else if (v.prio == 3) hipLaunchKernelGGL((vit_gemm_pp_kernel<256, EP, 1, 2>), dim3(nt), dim3(512), 0, st, a);
    else if (v.prio == 4) hipLaunchKernelGGL((vit_gemm_pp_kernel<256, EP, 1, 3>), dim3(nt), dim3(512), 0, st, a);
    else if (v.prio == 5) hipLaunchKernelGGL((vit_gemm_pp_kernel<256, EP, 1, 4>), dim3(nt), dim3(512), 0, st, a);
    else if (v.prio) hipLaunchKernelGGL((vit_gemm_pp_kernel<256, EP, 1>), dim3(nt), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((vit_gemm_pp_kernel<256, EP, 0>), dim3(nt), dim3(512), 0, st, a);
}
template <int EP>
static void launch(const Variant &v, GemmArgs a, hipStream_t st) {
    a.cb = v.cb;
    a.xpad = 1;
    if (v.kind == 0) {
        hipLaunchKernelGGL((vit_gemm_kernel<128, 128, 2, 4, EP, 0, 2>), dim3(a.N / 128, (a.M + 127) / 128), dim3(512), 0, st, a);
    } else if (v.kind == 1) {
        hipLaunchKernelGGL((vit_gemm_kernel<256, 256, 4, 4, EP, 0, 2>), dim3(a.N / 256, (a.M + 255) / 256), dim3(1024), 0, st, a);
    } else if constexpr (EP == EP_QKV) {       // as gemm_launch: q | k row-major, V transposed through the swapped-operand kernel
        GemmArgs qk = a, vv = a;
        qk.N = 2 * a.D;
        vv.N = a.D; vv.W = a.W + (size_t)2 * a.D * a.K; vv.bias = a.bias + 2 * a.D;
        launch_pp<EP_F16>(v, qk, st);
        launch_pp<EP_VT>(v, vv, st);
    } else {
        launch_pp<EP>(v, a, st);
    }
}
static void launch_ep(int ep, const Variant &v, const GemmArgs &a, hipStream_t st) {
    if (ep == EP_QKV) launch<EP_QKV>(v, a, st);
    else if (ep == EP_RESID) launch<EP_RESID>(v, a, st);
    else if (ep == EP_F16) launch<EP_F16>(v, a, st);
    else launch<EP_F16_GELU>(v, a, st);
}

int main(int argc, char **argv) {
    std::vector<int> frames;
    for (int i = 1; i < argc; ++i) frames.push_back(atoi(argv[i]));
    if (frames.empty()) frames = {8, 14, 16, 28, 32};
    const char *only = getenv("GEMM_PROBE_ONLY"), *only_gemm = getenv("GEMM_PROBE_GEMM");
    if (getenv("GEMM_PROBE_GRID")) g_grid = atoi(getenv("GEMM_PROBE_GRID"));
    const int iters = getenv("GEMM_PROBE_ITERS") ? atoi(getenv("GEMM_PROBE_ITERS")) : 20;
    const int S = 576, D = 1024, I = 4096, HD = 64;
    int maxB = 0;
    for (int b : frames) maxB = b > maxB ? b : maxB;
    const size_t maxM = (size_t)maxB * S;
    // operands: activations uniform [-1, 1), weights uniform * K^-1/2 (full-range random data, not zeros: cdna guide rule 25)
    std::vector<f16_t> hX(maxM * I), hW((size_t)I * I > (size_t)3 * D * D ? (size_t)I * D : (size_t)3 * D * D);
    for (auto &x : hX) x = h16(urand());
    for (auto &w : hW) w = h16(urand() * 0.03f);
    std::vector<float> hb(I);
    for (auto &b : hb) b = urand() * 0.1f;
    f16_t *X, *W, *out16[2], *vT[2];
    float *bias, *out32[2];
    CK(hipMalloc(&X, (maxM + 256) * I * 2));          // the ping-pong kernel reads whole 256-row tiles
    CK(hipMemset(X, 0, (maxM + 256) * I * 2));
    CK(hipMalloc(&W, hW.size() * 2));
    CK(hipMalloc(&bias, I * 4));
    for (int i = 0; i < 2; ++i) {
        CK(hipMalloc(&out16[i], maxM * I * 2));
        CK(hipMalloc(&vT[i], maxM * D * 2));
        CK(hipMalloc(&out32[i], maxM * D * 4));
    }
    CK(hipMemcpy(X, hX.data(), maxM * I * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), I * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    struct G { const char *name; int ep, N, K; } gemms[5] = {{"qkv", EP_QKV, 3 * D, D}, {"out", EP_RESID, D, D}, {"fc1", EP_F16_GELU, I, D}, {"fc2", EP_RESID, D, I},
                                                             {"fc1-nogelu", EP_F16, I, D}};
    std::vector<Variant> variants = {{"old128", 0, 128, 1, 0}, {"old256", 1, 256, 1, 0}, {"pp256", 2, 256, 0, 1}, {"pp256np", 2, 256, 0, 0},
                                     {"pp128", 2, 128, 0, 1},  {"pp256cb1", 2, 256, 1, 1}, {"pp256cb2", 2, 256, 2, 1}, {"pp256cb4", 2, 256, 4, 1},
                                     {"pp256cb8", 2, 256, 8, 1}, {"pp256noepi", 2, 256, 0, 2}, {"pp256noload", 2, 256, 0, 3}, {"pp256atomic", 2, 256, 0, 4}, {"pp256nostore", 2, 256, 0, 5}};
    for (int B : frames) {
        const int M = B * S;
        for (const G &g : gemms) {
            if (only_gemm && strcmp(only_gemm, g.name)) continue;
            GemmArgs a{};
            a.X = X; a.W = W; a.bias = bias; a.M = M; a.N = g.N; a.K = g.K; a.ldx = g.K; a.S = S; a.Sp = S; a.D = D; a.hd = HD;
            a.ldo = g.ep == EP_QKV ? 2 * D : g.N;
            const int tx = g.N / 256;
            int auto_cb = 1;
            while (auto_cb < 8 && tx % (auto_cb * 2) == 0 && (size_t)(g.N / auto_cb) * g.K * 2 > ((size_t)9 << 18)) auto_cb *= 2;
            const double flop = 2.0 * M * g.N * g.K;
            size_t out_bytes16 = (size_t)M * a.ldo * 2, out_bytes32 = (size_t)M * D * 4;
            bool have_ref = false;
            for (Variant v : variants) {
                if (only && strcmp(only, v.name)) continue;
                if (v.cb == 0) v.cb = auto_cb;
                if (v.kind == 2 && (tx % v.cb)) continue;
                if (!strncmp(v.name, "pp256cb", 7) && v.cb == auto_cb) continue;      // same as pp256
                const int slot = (v.kind == 0) ? 0 : 1;
                a.out16 = out16[slot]; a.outVT = vT[slot]; a.out32 = out32[slot];
                // correctness run on cleared outputs
                CK(hipMemsetAsync(out16[slot], 0, out_bytes16, st));
                CK(hipMemsetAsync(vT[slot], 0, (size_t)M * D * 2, st));
                CK(hipMemsetAsync(out32[slot], 0, out_bytes32, st));
                launch_ep(g.ep, v, a, st);
                CK(hipStreamSynchronize(st));
                CK(hipGetLastError());
                const char *verdict = "ref";
                if (v.kind == 0) have_ref = true;
                else if (v.prio == 2 || v.prio == 3 || v.prio == 5) verdict = "ablation";
                else if (v.prio == 4 && g.ep != EP_RESID) continue;
                else if (have_ref) {
                    std::vector<char> r0, r1;
                    auto cmp = [&](const void *p0, const void *p1, size_t n) {
                        r0.resize(n); r1.resize(n);
                        CK(hipMemcpy(r0.data(), p0, n, hipMemcpyDeviceToHost));
                        CK(hipMemcpy(r1.data(), p1, n, hipMemcpyDeviceToHost));
                        if (memcmp(r0.data(), r1.data(), n) == 0) return true;
                        if (g.ep != EP_RESID) {      // fp16 outputs: how many differ, and by how many fp16 ulps at most
                            const unsigned short *a16 = (const unsigned short *)r0.data(), *b16 = (const unsigned short *)r1.data();
                            size_t cnt = 0; int worst = 0;
                            for (size_t i = 0; i < n / 2; ++i)
                                if (a16[i] != b16[i]) { ++cnt; const int d = abs((int)(a16[i] & 0x7fff) - (int)(b16[i] & 0x7fff)); if (d > worst) worst = d; }
                            printf("    (%zu of %zu fp16 outputs differ, by at most %d in the last place)\n", cnt, n / 2, worst);
                        }
                        return false;
                    };
                    bool ok = true;
                    if (g.ep == EP_RESID) ok = cmp(out32[0], out32[1], out_bytes32);
                    else ok = cmp(out16[0], out16[1], out_bytes16);
                    if (g.ep == EP_QKV) ok = ok && cmp(vT[0], vT[1], (size_t)M * D * 2);
                    verdict = ok ? "bit-exact" : "MISMATCH";
                } else verdict = "unchecked";
                for (int i = 0; i < 3; ++i) launch_ep(g.ep, v, a, st);
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i) launch_ep(g.ep, v, a, st);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / iters;
                if (v.kind == 0) {       // the timing launches accumulated into the fp32 residual stream: restore the one-launch reference
                    CK(hipMemsetAsync(out32[0], 0, out_bytes32, st));
                    launch_ep(g.ep, v, a, st);
                    CK(hipStreamSynchronize(st));
                }
                printf("B=%2d %s M %5d N %4d K %4d %-9s cb %d: %8.1f us %7.0f TFLOP/s  %s\n", B, g.name, M, g.N, g.K, v.name, v.cb, us, flop / us / 1e6, verdict);
                fflush(stdout);
            }
        }
    }
    return 0;
}
