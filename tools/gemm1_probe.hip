// gemm1_probe.hip — the SigLIP-L GEMMs of ONE frame (or two, three) through the engine's own kernels, launch sequence against launch sequence:
//   "cur"   the 64 x 64 tiles the tower ran through round 5 (csrc/vit_gemm.inc::gemm_launch / gemm_launch_slab exactly as vit.hip called them),
//   "tall"  the tall-tile kernel of csrc/vit_tall.inc (144 x 64, one workgroup per CU, software-pipelined K loop),
//   each alone and with the LayerNorm launch the tower runs in front of it (q|k|v, fc1) / behind it (out-proj, fc2: direct epilogue + LayerNorm, or
//   split-K slabs + the LayerNorm that reduces them).
// Per sequence: microseconds (HIP events over a chain of `iters` repetitions on one stream; every repetition reads a DIFFERENT copy of the weights out of a
// 384 MiB arena — in the tower every layer has its own weights, nothing a launch reads of W is in L2 or the 256 MiB memory-side cache) and a bit-exact
// check of the new form against the old one.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm1_probe.hip -o tools/_bin/gemm1_probe ;  tools/_bin/gemm1_probe [frames ...]   (default 1 2)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "../videollm-online_amd/csrc/common.cuh"
#include "../videollm-online_amd/csrc/vit_gemm.inc"
#include "../videollm-online_amd/csrc/vit_ln.inc"
#include "../videollm-online_amd/csrc/vit_tall.inc"

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__);      \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float urand() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (float)((rng_state >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f;
}
static f16_t h16(float f) { _Float16 t = (_Float16)f; f16_t r; memcpy(&r, &t, 2); return r; }

static hipStream_t st;
static hipEvent_t e0, e1;
static int iters = 48;

// time `seq(i)` (i = repetition index, picks the weight copy) over `iters` repetitions
static double time_us(const std::function<void(int)> &seq) {
    for (int i = 0; i < 4; ++i) seq(i);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) seq(i + 4);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / iters;
}
// bit comparison; on a mismatch prints how many 16-bit (fp16 outputs) or 32-bit (fp32 residual stream) words differ and by how much at most
static bool same(const void *p0, const void *p1, size_t n, bool f32 = false) {
    std::vector<char> r0(n), r1(n);
    CK(hipMemcpy(r0.data(), p0, n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r1.data(), p1, n, hipMemcpyDeviceToHost));
    if (memcmp(r0.data(), r1.data(), n) == 0) return true;
    size_t cnt = 0;
    double worst = 0;
    if (f32) {
        const float *a = (const float *)r0.data(), *b = (const float *)r1.data();
        for (size_t i = 0; i < n / 4; ++i) if (a[i] != b[i]) { ++cnt; worst = std::max(worst, (double)fabsf(a[i] - b[i])); }
        printf("      (%zu of %zu fp32 words differ, largest |difference| %.3g)\n", cnt, n / 4, worst);
    } else {
        const unsigned short *a = (const unsigned short *)r0.data(), *b = (const unsigned short *)r1.data();
        for (size_t i = 0; i < n / 2; ++i) if (a[i] != b[i]) { ++cnt; worst = std::max(worst, (double)abs((int)(a[i] & 0x7fff) - (int)(b[i] & 0x7fff))); }
        printf("      (%zu of %zu fp16 words differ, by at most %.0f in the last place)\n", cnt, n / 2, worst);
    }
    return false;
}

// phase boundaries of one launch of the tall kernel (VIT_STAMP, s_memrealtime = 100 MHz): mean and latest workgroup, relative to the first workgroup's start
static unsigned long long *g_stamps = nullptr;
static void stamp_report(const char *what, int blocks, const std::function<void()> &launch) {
    CK(hipMemsetAsync(g_stamps, 0, (size_t)blocks * 32, st));
    launch();
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h((size_t)blocks * 4);
    CK(hipMemcpy(h.data(), g_stamps, (size_t)blocks * 32, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < blocks; ++b) if (h[(size_t)b * 4]) t0 = std::min(t0, h[(size_t)b * 4]);
    printf("      %s, us after the first workgroup's start (mean / latest workgroup):", what);
    const char *names[4] = {"start", "first K tile landed", "K loop done", "end"};
    for (int k = 0; k < 4; ++k) {
        double sum = 0, mx = 0; int n = 0;
        for (int b = 0; b < blocks; ++b) { const unsigned long long v = h[(size_t)b * 4 + k]; if (v) { const double us = (double)(v - t0) / 100.0; sum += us; mx = std::max(mx, us); ++n; } }
        if (n) printf(" %s %.2f / %.2f;", names[k], sum / n, mx);
    }
    printf("\n");
}

int main(int argc, char **argv) {
    std::vector<int> frames;
    for (int i = 1; i < argc; ++i) frames.push_back(atoi(argv[i]));
    if (frames.empty()) frames = {1, 2};
    if (getenv("GEMM_PROBE_ITERS")) iters = atoi(getenv("GEMM_PROBE_ITERS"));
    const char *only = getenv("GEMM_PROBE_GEMM");
    const bool warm_w = getenv("GEMM_PROBE_WARMW") && atoi(getenv("GEMM_PROBE_WARMW"));      // every launch reads the SAME weights (L2 / memory-side cache warm)
    const int S = 576, D = 1024, I = 4096, HD = 64;
    int maxB = 0;
    for (int b : frames) maxB = std::max(maxB, b);
    const size_t maxM = (size_t)maxB * S;
    std::vector<f16_t> hX(maxM * I), hW((size_t)I * D);
    for (auto &x : hX) x = h16(urand());
    for (auto &w : hW) w = h16(urand() * 0.03f);
    std::vector<float> hb(I), hh(maxM * D), hg(D), hbeta(D);
    for (auto &b : hb) b = urand() * 0.1f;
    for (auto &v : hh) v = urand() * 2.0f;
    for (auto &v : hg) v = 1.0f + 0.1f * urand();
    for (auto &v : hbeta) v = 0.1f * urand();
    const size_t arena = (size_t)384 << 20, wbytes = hW.size() * 2;
    f16_t *X, *Warena, *x16[2], *out16[2], *vT[2];
    float *bias, *h[3], *slab, *lnw, *lnb;
    CK(hipMalloc(&X, (maxM + 256) * I * 2));
    CK(hipMemset(X, 0, (maxM + 256) * I * 2));
    CK(hipMalloc(&Warena, arena));
    CK(hipMalloc(&bias, I * 4));
    CK(hipMalloc(&lnw, D * 4));
    CK(hipMalloc(&lnb, D * 4));
    CK(hipMalloc(&slab, 4 * maxM * D * 4));
    CK(hipMalloc(&g_stamps, 1024 * 32));
    for (int i = 0; i < 3; ++i) CK(hipMalloc(&h[i], maxM * D * 4));
    for (int i = 0; i < 2; ++i) {
        CK(hipMalloc(&x16[i], (maxM + 256) * D * 2));
        CK(hipMemset(x16[i], 0, (maxM + 256) * D * 2));
        CK(hipMalloc(&out16[i], maxM * I * 2));
        CK(hipMalloc(&vT[i], maxM * D * 2));
    }
    CK(hipMemcpy(X, hX.data(), maxM * I * 2, hipMemcpyHostToDevice));
    // every slot of every shape holds the same matrix: copies at 2 MiB granularity (the smallest W) of the first 2 MiB ... no: whole copies of hW at the
    // LARGEST size; smaller shapes use slot strides of their own size, so their slots differ in content — the checks below always use slot 0
    for (size_t o = 0; o + wbytes <= arena; o += wbytes) CK(hipMemcpy((char *)Warena + o, hW.data(), wbytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), I * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(lnw, hg.data(), D * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(lnb, hbeta.data(), D * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(h[2], hh.data(), maxM * D * 4, hipMemcpyHostToDevice));          // pristine residual stream
    CK(hipStreamCreate(&st));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto Wcopy = [&](int i, size_t wsz) { return (const f16_t *)((const char *)Warena + (size_t)(warm_w ? 0 : i % (arena / wsz)) * wsz); };
    auto reset_h = [&](int k, int M) { CK(hipMemcpyAsync(h[k], h[2], (size_t)M * D * 4, hipMemcpyDeviceToDevice, st)); };
    auto report = [&](int B, const char *gemm, const char *what, double us, double flop, const char *verdict) {
        printf("B=%d %-4s %-36s: %7.2f us %6.0f TFLOP/s  %s\n", B, gemm, what, us, flop / us / 1e6, verdict);
        fflush(stdout);
    };

    for (int B : frames) {
        const int M = B * S;
        auto ln_plain = [&](float *hh_, f16_t *o16) {
            hipLaunchKernelGGL((vit_layernorm_kernel<false>), dim3((M + 3) / 4), dim3(256), 0, st, hh_, (const float *)lnw, (const float *)lnb, o16, (float *)nullptr, M, D, 1e-6f,
                               (const float *)nullptr, 0, (size_t)0, (const float *)nullptr);
        };
        // ---------------- q|k|v and fc1
        struct G { const char *name; int ep, N; } lg[2] = {{"qkv", EP_QKV, 3 * D}, {"fc1", EP_F16_GELU, I}};
        for (const G &g : lg) {
            if (only && strcmp(only, g.name)) continue;
            const size_t wsz = (size_t)g.N * D * 2;
            const double flop = 2.0 * M * g.N * D;
            GemmArgs a{};
            a.bias = bias; a.M = M; a.N = g.N; a.K = D; a.ldx = D; a.S = S; a.Sp = S; a.D = D; a.hd = HD; a.xpad = 1;
            a.ldo = g.ep == EP_QKV ? 2 * D : g.N;
            auto gemm_cur = [&](int i, int slot) {
                GemmArgs b = a; b.X = x16[slot]; b.W = Wcopy(i, wsz); b.out16 = out16[slot]; b.outVT = vT[slot];
                CK(g.ep == EP_QKV ? gemm_launch<EP_QKV>(b, st) : gemm_launch<EP_F16_GELU>(b, st));
            };
            auto gemm_tall = [&](int i, int slot) {
                GemmArgs b = a; b.X = x16[slot]; b.W = Wcopy(i, wsz); b.out16 = out16[slot]; b.outVT = vT[slot];
                CK(g.ep == EP_QKV ? (gemm_launch_tall<EP_QKV>(b, 1, st)) : (gemm_launch_tall<EP_F16_GELU>(b, 1, st)));
            };
            const double t_cur = time_us([&](int i) { gemm_cur(i, 0); });
            const double t_tall = time_us([&](int i) { gemm_tall(i, 1); });
            const double t_lcur = time_us([&](int i) { ln_plain(h[2], x16[0]); gemm_cur(i, 0); });
            const double t_ltall = time_us([&](int i) { ln_plain(h[2], x16[1]); gemm_tall(i, 1); });
            // the check LAST, both on weight slot 0 (the timed chains left other slots' results in the buffers)
            CK(hipMemsetAsync(out16[0], 0, (size_t)M * a.ldo * 2, st)); CK(hipMemsetAsync(vT[0], 0, (size_t)M * D * 2, st));
            CK(hipMemsetAsync(out16[1], 0, (size_t)M * a.ldo * 2, st)); CK(hipMemsetAsync(vT[1], 0, (size_t)M * D * 2, st));
            ln_plain(h[2], x16[0]); gemm_cur(0, 0);
            ln_plain(h[2], x16[1]); gemm_tall(0, 1);
            CK(hipStreamSynchronize(st));
            bool ok = same(out16[0], out16[1], (size_t)M * a.ldo * 2);
            if (g.ep == EP_QKV) ok = same(vT[0], vT[1], (size_t)M * D * 2) && ok;
            const char *v = ok ? "bit-exact" : "MISMATCH";
            report(B, g.name, "cur (64x64) GEMM alone", t_cur, flop, "ref");
            report(B, g.name, "tall GEMM alone", t_tall, flop, v);
            stamp_report("tall GEMM", ((M + 143) / 144) * (g.N / 64), [&]() {
                GemmArgs b = a; b.X = x16[1]; b.W = Wcopy(9, wsz); b.out16 = out16[1]; b.outVT = vT[1]; b.stamps = g_stamps;
                CK(g.ep == EP_QKV ? (gemm_launch_tall<EP_QKV>(b, 1, st)) : (gemm_launch_tall<EP_F16_GELU>(b, 1, st)));
            });
            report(B, g.name, "LayerNorm launch + cur GEMM", t_lcur, flop, "ref");
            report(B, g.name, "LayerNorm launch + tall GEMM", t_ltall, flop, v);
        }
        // ---------------- out-proj and fc2: GEMM into the residual stream (+ the LayerNorm behind it)
        struct R { const char *name; int K; } rg[2] = {{"out", D}, {"fc2", I}};
        for (const R &g : rg) {
            if (only && strcmp(only, g.name)) continue;
            const size_t wsz = (size_t)D * g.K * 2;
            const double flop = 2.0 * M * D * g.K;
            GemmArgs a{};
            a.X = X; a.bias = bias; a.M = M; a.N = D; a.K = g.K; a.ldx = g.K; a.xpad = 1;
            auto direct_cur = [&](int i, int k) { GemmArgs b = a; b.W = Wcopy(i, wsz); b.out32 = h[k]; CK(gemm_launch<EP_RESID>(b, st)); };
            auto direct_tall = [&](int i, int k) { GemmArgs b = a; b.W = Wcopy(i, wsz); b.out32 = h[k]; CK((gemm_launch_tall<EP_RESID>(b, 1, st))); };
            auto slab_cur = [&](int i, int ks) { GemmArgs b = a; b.W = Wcopy(i, wsz); b.out32 = slab; b.ldo = M; CK(gemm_launch_slab(b, ks, st)); };
            auto slab_tall = [&](int i, int ks) { GemmArgs b = a; b.W = Wcopy(i, wsz); b.out32 = slab; b.ldo = M; CK((gemm_launch_tall<EP_SLAB>(b, ks, st))); };
            auto ln_red = [&](int k, int ks) {
                hipLaunchKernelGGL((vit_layernorm_kernel<true>), dim3((M + 3) / 4), dim3(256), 0, st, h[k], (const float *)lnw, (const float *)lnb, x16[k], (float *)nullptr, M, D, 1e-6f,
                                   (const float *)slab, ks, (size_t)M * D, (const float *)bias);
            };
            report(B, g.name, "cur direct (EP_RESID)", time_us([&](int i) { direct_cur(i, 0); }), flop, "ref");
            const double t_dt = time_us([&](int i) { direct_tall(i, 1); });
            const double t_dcl = time_us([&](int i) { direct_cur(i, 0); ln_plain(h[0], x16[0]); });
            const double t_dtl = time_us([&](int i) { direct_tall(i, 1); ln_plain(h[1], x16[1]); });
            reset_h(0, M); direct_cur(0, 0);
            reset_h(1, M); direct_tall(0, 1);
            CK(hipStreamSynchronize(st));
            const char *vd = same(h[0], h[1], (size_t)M * D * 4, true) ? "bit-exact" : "MISMATCH";
            report(B, g.name, "tall direct (EP_RESID)", t_dt, flop, vd);
            stamp_report("tall direct", ((M + 143) / 144) * (D / 64), [&]() { GemmArgs b = a; b.W = Wcopy(9, wsz); b.out32 = h[1]; b.stamps = g_stamps; CK((gemm_launch_tall<EP_RESID>(b, 1, st))); });
            report(B, g.name, "cur direct + LayerNorm launch", t_dcl, flop, "ref");
            report(B, g.name, "tall direct + LayerNorm launch", t_dtl, flop, vd);
            for (int ks : {2, 4}) {
                char nm[96];
                if (g.K / 64 / ks < 3) continue;
                const double t_sc = time_us([&](int i) { slab_cur(i, ks); ln_red(0, ks); });
                const double t_st = time_us([&](int i) { slab_tall(i, ks); ln_red(1, ks); });
                const double t_st0 = time_us([&](int i) { slab_tall(i, ks); });
                reset_h(0, M); slab_cur(0, ks); ln_red(0, ks);
                reset_h(1, M); slab_tall(0, ks); ln_red(1, ks);
                CK(hipStreamSynchronize(st));
                const char *vs = same(h[0], h[1], (size_t)M * D * 4, true) && same(x16[0], x16[1], (size_t)M * D * 2) ? "bit-exact" : "MISMATCH";
                snprintf(nm, sizeof nm, "cur slab k%d + LayerNorm<reduce>", ks);
                report(B, g.name, nm, t_sc, flop, "ref");
                snprintf(nm, sizeof nm, "tall slab k%d + LayerNorm<reduce>", ks);
                report(B, g.name, nm, t_st, flop, vs);
                snprintf(nm, sizeof nm, "tall slab k%d alone", ks);
                report(B, g.name, nm, t_st0, flop, vs);
                stamp_report("tall slab", ((M + 143) / 144) * (D / 64) * ks, [&]() { GemmArgs b = a; b.W = Wcopy(9, wsz); b.out32 = slab; b.ldo = M; b.stamps = g_stamps; CK((gemm_launch_tall<EP_SLAB>(b, ks, st))); });
            }
        }
    }
    return 0;
}
