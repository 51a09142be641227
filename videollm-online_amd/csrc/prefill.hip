// prefill.hip — the Llama projections for a BLOCK of up to 64 new tokens per weight pass.
//
// The streaming step (gemv.hip) moves every weight byte once per <= 16 tokens, which is the right shape for the live
// path (n = 11 frame steps, n = 1 decode) but makes a long teacher-forced input — stream_evaluate's whole-dialogue
// forward, models/modeling_live.py:67, ~13 k tokens on the reference's data — pay one 15 GB weight pass per 16 tokens.
// This kernel streams the SAME packed weight image (A-operand fragments, gemv.hip) once per 64 tokens: each weight
// fragment feeds four MFMAs (one per 16-token tile) instead of one, so the pass stays HBM-bound (64 FLOP per weight
// byte vs the ~310 ridge) while the tokens per byte quadruple.
//
//   D[ncol][token] = sum_k Wfrag[ncol][k] * xfrag[k][token]      (W as A, x as B, v_mfma_f32_16x16x32_bf16)
//
// Work split = gemv16_kernel's: persistent blocks stride over groups of two column tiles (rotary pairs for q/k/v,
// 8 gate + 8 up rows per tile for SwiGLU), the block's waves split K, a wave walks its K range in chunks of KF
// fragments, partial tiles meet in LDS and the epilogue runs on the reduced sums.  Differences: 4 token tiles per
// weight fragment; the activation fragments of a chunk (4 x KF) are re-read from L2 per chunk (x is <= 1.8 MB and
// shared by every block) and therefore live in a fragment-packed layout ("packed-64", llm_ops.h) written by their
// producers — row-major rows cost 16 half-used cache lines per fragment and made the kernel L2-line-bound; no norm-on-load / split-K partial variants (the block path runs the row-parallel
// add_rmsnorm kernel instead, its cost is amortised over 64 tokens).
//
// WQ = 1 streams the fp8 e4m3 image (gemv.hip: one 16-byte register = two consecutive fragments, expanded to bf16 exactly in
// registers; per-output-channel scales on the reduced fp32 sums): one expansion feeds the four token tiles.
#include <stdlib.h>

#include "common.cuh"
#include "prefill.h"

VLO_DEV float silu_bf16_p(float g) { return rbf(g / (1.0f + __expf(-g))); }     // F.silu on a bf16 tensor
VLO_DEV float4 f4add_p(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int KF, int EPI, int WQ>
__global__ __launch_bounds__(512) void gemm64_kernel(GemvArgs a) {
    constexpr int MT = VLO_BLOCK_TOKENS / 16, CTG = 2;
    constexpr int WRN = WQ ? KF / 2 : KF;                                  // weight registers per tile and K chunk (fp8: two fragments each)
    static_assert(!WQ || (KF % 2) == 0, "fp8 image: fragments come in pairs");
    extern __shared__ __attribute__((aligned(16))) float4 red[];          // [NW][CTG][MT][64] float4
    const int NW = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int KFtot = a.K >> 5;
    const int kfw0 = w * a.KC * KF;                                        // first fragment of this wave's K range
    const int mt_live = (a.n_rows + 15) >> 4;                              // token tiles that hold real rows

    const int hd = a.kv.head_dim;
    const int tph = (EPI == EPI_ROPE) ? hd / 16 : 2;
    const int hp = tph / 2;
    const bool single = (EPI != EPI_ROPE && a.CT == 1);
    const int ngroups = (EPI == EPI_ROPE) ? a.NT / 2 : (single ? a.NT : (a.NT + 1) / 2);
    auto tile_a = [&](int g) { return (EPI == EPI_ROPE) ? (g / hp) * tph + (g % hp) : (single ? g : 2 * g); };
    auto tile_b = [&](int g) { return (EPI == EPI_ROPE) ? (g / hp) * tph + (g % hp) + hp : (single ? a.NT : 2 * g + 1); };

    const frag_ab *wbase = reinterpret_cast<const frag_ab *>(a.Wp) + (size_t)(WQ ? kfw0 / 2 : kfw0) * 64 + lane;
    const size_t tile_stride = (size_t)(WQ ? KFtot / 2 : KFtot) * 64;
    auto item_ptr = [&](int tile, int c) { return wbase + (size_t)tile * tile_stride + (size_t)c * WRN * 64; };

    // Weight fragments of BOTH tiles of the current K chunk sit in registers (2 x KF KiB per wave in flight) and each is
    // refilled for the next chunk right after the MFMAs that consumed it; the four activation fragments of step kf+1
    // are fetched (L2) while step kf computes.
    frag_ab wrA[WRN], wrB[WRN];
    frag_ab xc[MT], xn[MT];
    const frag_ab *xbase = reinterpret_cast<const frag_ab *>(a.x) + (size_t)kfw0 * MT * 64 + lane;   // packed-64 (llm_ops.h)
    auto load_x = [&](int step, frag_ab (&dst)[MT]) {                      // step = c * KF + kf inside this wave's K range
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if (mt < mt_live) dst[mt] = xbase[((size_t)step * MT + mt) * 64];
    };
    int g = blockIdx.x;
    if (g < ngroups) {
        const frag_ab *pa = item_ptr(tile_a(g), 0);
        const int tb = tile_b(g);
#pragma unroll
        for (int i = 0; i < WRN; ++i) wrA[i] = __builtin_nontemporal_load(pa + i * 64);
        if (tb < a.NT) {
            const frag_ab *pb = item_ptr(tb, 0);
#pragma unroll
            for (int i = 0; i < WRN; ++i) wrB[i] = __builtin_nontemporal_load(pb + i * 64);
        }
        load_x(0, xc);
    }

    for (; g < ngroups; g += gridDim.x) {
        const int tA = tile_a(g), tB = tile_b(g);
        const bool hasB = tB < a.NT;
        const int gn = g + gridDim.x;
        f32x4 acc[CTG][MT];
#pragma unroll
        for (int t = 0; t < CTG; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < a.KC; ++c) {
            const bool last_c = c + 1 == a.KC;
            // where this chunk's registers are refilled from: the next chunk of this group, or chunk 0 of the next group
            const frag_ab *na = !last_c ? item_ptr(tA, c + 1) : (gn < ngroups ? item_ptr(tile_a(gn), 0) : nullptr);
            const int tBn = !last_c ? tB : (gn < ngroups ? tile_b(gn) : a.NT);
            const frag_ab *nb = tBn < a.NT ? item_ptr(tBn, last_c ? 0 : c + 1) : nullptr;
            frag_ab eA[2], eB[2];                                          // fp8: the expanded fragment pair of the current register
#pragma unroll
            for (int kf = 0; kf < KF; ++kf) {
                // x of the next step (wraps to step 0 — same K range — when the next group starts)
                const int nstep = (last_c && kf == KF - 1) ? 0 : c * KF + kf + 1;
                load_x(nstep, xn);
                if (WQ && !(kf & 1)) {                                     // expand, then the register is free for its refill
                    fp8x16_to_bf16(wrA[kf >> 1], eA[0], eA[1]);
                    if (na) wrA[kf >> 1] = __builtin_nontemporal_load(na + (kf >> 1) * 64);
                    if (hasB) fp8x16_to_bf16(wrB[kf >> 1], eB[0], eB[1]);
                    if (nb) wrB[kf >> 1] = __builtin_nontemporal_load(nb + (kf >> 1) * 64);
                }
                const frag_ab fa = WQ ? eA[kf & 1] : wrA[kf];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (mt < mt_live) acc[0][mt] = mfma_bf16(fa, xc[mt], acc[0][mt]);
                if (!WQ && na) wrA[kf] = __builtin_nontemporal_load(na + kf * 64);
                if (hasB) {
                    const frag_ab fb = WQ ? eB[kf & 1] : wrB[kf];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        if (mt < mt_live) acc[1][mt] = mfma_bf16(fb, xc[mt], acc[1][mt]);
                }
                if (!WQ && nb) wrB[kf] = __builtin_nontemporal_load(nb + kf * 64);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xc[mt] = xn[mt];
            }
        }
#pragma unroll
        for (int t = 0; t < CTG; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                red[((size_t)(w * CTG + t) * MT + mt) * 64 + lane] = make_float4(acc[t][mt][0], acc[t][mt][1], acc[t][mt][2], acc[t][mt][3]);
        __syncthreads();

        auto reduced = [&](int ct, int mt, int l) {
            float4 s = make_float4(0, 0, 0, 0);
            for (int ww = 0; ww < NW; ++ww) s = f4add_p(s, red[((size_t)(ww * CTG + ct) * MT + mt) * 64 + l]);
            if (WQ) {                                                      // per-output-channel weight scales, packed row order (gemv.h)
                const float4 sc = *reinterpret_cast<const float4 *>(a.wscale + (ct ? tB : tA) * 16 + (l >> 4) * 4);
                s = make_float4(s.x * sc.x, s.y * sc.y, s.z * sc.z, s.w * sc.w);
            }
            return s;
        };
        if (EPI == EPI_ROPE) {
            // a lane needs both tiles of the rotary pair: items = (token tile, lane)
            for (int t = threadIdx.x; t < MT * 64; t += blockDim.x) {
                const int mt = t >> 6, l = t & 63;
                const int m = mt * 16 + (l & 15);
                if (m >= a.n_rows) continue;
                const float4 sA = reduced(0, mt, l), sB = reduced(1, mt, l);
                const float va[4] = {sA.x, sA.y, sA.z, sA.w}, vb[4] = {sB.x, sB.y, sB.z, sB.w};
                const int half = hd >> 1, nh = a.num_heads, nkv = a.kv.num_kv_heads;
                const int head = g / hp, i = (g % hp) * 16 + (l >> 4) * 4;           // column inside the head, < half
                const long long pos = a.pos0 + m;
                const int page = a.kv.page_table[pos / VLO_PAGE_TOKENS];
                const int tok = (int)(pos % VLO_PAGE_TOKENS);
                if (head < nh + nkv) {
                    bf16_t *dst = (head < nh)
                        ? a.out_bf16 + (size_t)m * nh * hd + (size_t)head * hd
                        : a.kv.k_pool + (size_t)a.layer * a.kv.layer_stride + (size_t)page * a.kv.page_elems +
                              ((size_t)(head - nh) * VLO_PAGE_TOKENS + tok) * hd;
                    const ushort4 c4 = *reinterpret_cast<const ushort4 *>(a.cos_tab + pos * half + i);
                    const ushort4 s4 = *reinterpret_cast<const ushort4 *>(a.sin_tab + pos * half + i);
                    const bf16_t cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
                    bf16_t lo[4], hi[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x1 = rbf(va[r]), x2 = rbf(vb[r]);              // projection output is bf16
                        const float cs = bf2f(cc[r]), sn = bf2f(ss[r]);
                        lo[r] = f2bf(rbf(x1 * cs) + rbf(-x2 * sn));                // q*cos + rotate_half(q)*sin, bf16 at every op
                        hi[r] = f2bf(rbf(x2 * cs) + rbf(x1 * sn));
                    }
                    *reinterpret_cast<ushort4 *>(dst + i) = *reinterpret_cast<const ushort4 *>(lo);
                    *reinterpret_cast<ushort4 *>(dst + half + i) = *reinterpret_cast<const ushort4 *>(hi);
                } else {
                    bf16_t *dst = a.kv.vt_pool + (size_t)a.layer * a.kv.layer_stride + (size_t)page * a.kv.page_elems +
                                  ((size_t)(head - nh - nkv) * hd) * VLO_PAGE_TOKENS + tok;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dst[(size_t)(i + r) * VLO_PAGE_TOKENS] = f2bf(va[r]);
                        dst[(size_t)(half + i + r) * VLO_PAGE_TOKENS] = f2bf(vb[r]);
                    }
                }
            }
        } else {
            for (int t = threadIdx.x; t < CTG * MT * 64; t += blockDim.x) {
                const int ct = t / (MT * 64), mt = (t >> 6) % MT, l = t & 63;
                const int tile = ct ? tB : tA;
                const int m = mt * 16 + (l & 15);
                if (tile >= a.NT || m >= a.n_rows) continue;
                const int col = tile * 16 + (l >> 4) * 4;
                if (EPI == EPI_SWIGLU) {
                    // tile = 8 gate rows (lanes 0..31) + the 8 up rows of the same columns (lanes 32..63)
                    if (l >= 32) continue;
                    const float4 gsum = reduced(ct, mt, l), usum = reduced(ct, mt, l + 32);
                    const float gv[4] = {gsum.x, gsum.y, gsum.z, gsum.w}, uv[4] = {usum.x, usum.y, usum.z, usum.w};
                    bf16_t o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = f2bf(silu_bf16_p(rbf(gv[r])) * rbf(uv[r]));
                    // act feeds the down projection of the block path: packed-64 (4 consecutive k = half a 16-byte unit)
                    *reinterpret_cast<ushort4 *>(a.out_bf16 + vlo_pack64_elem(m, tile * 8 + (l >> 4) * 4)) = *reinterpret_cast<const ushort4 *>(o);
                } else if (EPI == EPI_RESID) {
                    const float4 s = reduced(ct, mt, l);
                    bf16_t *hp4 = a.h + (size_t)m * a.ldo + col;
                    const ushort4 hv = *reinterpret_cast<const ushort4 *>(hp4);
                    const bf16_t hh[4] = {hv.x, hv.y, hv.z, hv.w};
                    const float sv[4] = {s.x, s.y, s.z, s.w};
                    bf16_t o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = f2bf(rbf(bf2f(hh[r]) + rbf(sv[r])));   // Linear output -> bf16, then the bf16 residual add
                    *reinterpret_cast<ushort4 *>(hp4) = *reinterpret_cast<const ushort4 *>(o);
                } else {                                                   // EPI_BF16
                    if (col >= a.N_valid) continue;                        // N padded to 16 at pack time; N_valid % 4 == 0
                    const float4 s = reduced(ct, mt, l);
                    ushort4 o;
                    o.x = f2bf(s.x); o.y = f2bf(s.y); o.z = f2bf(s.z); o.w = f2bf(s.w);
                    *reinterpret_cast<ushort4 *>(a.out_bf16 + (size_t)m * a.ldo + col) = o;
                }
            }
        }
        __syncthreads();                                                   // `red` is rewritten by the next group
    }
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
int gemm64_plan(int K, Gemm64Plan *p, bool even_kf) {
    if (K <= 0 || (K & 31)) return -1;
    const int KFtot = K >> 5;
    // 8-wave blocks, one per CU, each wave keeping 2 x KF KiB of weight loads in flight (KF = 8: 128 KiB per CU, what the
    // 16-row GEMV keeps); the first try (4 KiB per wave, <= 12 waves per CU) was latency-bound at ~2 TB/s
    static const int nws[] = {8, 4, 2, 1}, kfs[] = {8, 4, 2, 1};
    for (int nw : nws)
        for (int kf : kfs)
            if (KFtot % (nw * kf) == 0 && !(even_kf && (kf & 1))) {      // fp8 image: fragments are stored in pairs
                p->NW = nw; p->KF = kf; p->KC = KFtot / (nw * kf);
                return 0;
            }
    return -1;
}

template <int KF, int WQ>
static hipError_t launch64(const GemvArgs &a, int epi, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
#define VLO_GO64(EP)                                                                      \
    do {                                                                                  \
        static bool attr = false;                                                         \
        if (!attr) {                                                                      \
            hipFuncSetAttribute((const void *)gemm64_kernel<KF, EP, WQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
            attr = true;                                                                  \
        }                                                                                 \
        hipLaunchKernelGGL((gemm64_kernel<KF, EP, WQ>), grid, block, lds, st, a);             \
        return hipGetLastError();                                                         \
    } while (0)
    if (epi == EPI_ROPE) VLO_GO64(EPI_ROPE);
    if (epi == EPI_SWIGLU) VLO_GO64(EPI_SWIGLU);
    if (epi == EPI_RESID) VLO_GO64(EPI_RESID);
    if (epi == EPI_BF16) VLO_GO64(EPI_BF16);
#undef VLO_GO64
    return hipErrorInvalidValue;
}

hipError_t gemm64_launch(GemvArgs a, const Gemm64Plan &p, int epi, hipStream_t st) {
    if (a.n_rows <= 0 || a.n_rows > VLO_BLOCK_TOKENS) return hipErrorInvalidValue;
    if (epi == EPI_ROPE && ((a.NT & 1) || (a.kv.head_dim != 64 && a.kv.head_dim != 128))) return hipErrorInvalidValue;
    a.KC = p.KC;
    // single column tiles when pairs would leave most CUs without work (o_proj / down_proj: 256 tiles)
    a.CT = (epi != EPI_ROPE && (a.NT + 1) / 2 < 256) ? 1 : 2;
    const int ngroups = (epi == EPI_ROPE) ? a.NT / 2 : (a.CT == 1 ? a.NT : (a.NT + 1) / 2);
    int gx = ngroups < 256 ? ngroups : 256;                                // one resident 8-wave block per CU
    const int per = (ngroups + gx - 1) / gx;
    gx = (ngroups + per - 1) / per;
    const size_t lds = (size_t)p.NW * 2 * (VLO_BLOCK_TOKENS / 16) * 64 * sizeof(float4);
    dim3 grid(gx), block(p.NW * 64);
    if (a.wq) {                                  // fp8 image: fragment pairs
        if (!a.wscale || (a.K & 63)) return hipErrorInvalidValue;
        switch (p.KF) {
            case 8: return launch64<8, 1>(a, epi, grid, block, lds, st);
            case 4: return launch64<4, 1>(a, epi, grid, block, lds, st);
            case 2: return launch64<2, 1>(a, epi, grid, block, lds, st);
        }
        return hipErrorInvalidValue;
    }
    switch (p.KF) {
        case 8: return launch64<8, 0>(a, epi, grid, block, lds, st);
        case 4: return launch64<4, 0>(a, epi, grid, block, lds, st);
        case 2: return launch64<2, 0>(a, epi, grid, block, lds, st);
        case 1: return launch64<1, 0>(a, epi, grid, block, lds, st);
    }
    return hipErrorInvalidValue;
}


// ------------------------------------------------------------------------------------
// long inputs: the projections as ping-pong GEMMs over the packed weight image (prefill.h)
// ------------------------------------------------------------------------------------
#include <algorithm>

#define VLO_GEMM_KERNELS_ONLY
#include "vit_gemm.inc"

// one 16-byte register of the fp8 image = the 8 + 8 codes of fragments 2 kf2 and 2 kf2 + 1 of a lane -> two 16-byte bf16 fragments
__global__ __launch_bounds__(256) void expand_fp8_image_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n_regs, int KF2tot) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_regs; i += (size_t)gridDim.x * blockDim.x) {
        const size_t lane = i & 63, t = i >> 6, kf2 = t % KF2tot, tile = t / KF2tot;
        frag_ab f0, f1;
        fp8x16_to_bf16(__builtin_bit_cast(frag_ab, src[i]), f0, f1);
        uint4 *d = dst + ((tile * KF2tot + kf2) * 2) * 64 + lane;
        d[0] = __builtin_bit_cast(uint4, f0);
        d[64] = __builtin_bit_cast(uint4, f1);
    }
}
hipError_t expand_fp8_image_launch(const void *Wp8, void *Wp_bf16, int NT, int K, hipStream_t st) {
    if (!Wp8 || !Wp_bf16 || NT <= 0 || (K & 63)) return hipErrorInvalidValue;
    const size_t n_regs = (size_t)NT * (K >> 6) * 64;
    const int blocks = (int)std::min<size_t>((n_regs + 255) / 256, 4096);
    hipLaunchKernelGGL(expand_fp8_image_kernel, dim3(blocks), dim3(256), 0, st, (const uint4 *)Wp8, (uint4 *)Wp_bf16, n_regs, K >> 6);
    return hipGetLastError();
}

// tile height: 256 rows once such tiles fill the chip, else 128 (twice the tiles, half the work each).  Round 6 tried 128-row tiles where the 256-row
// ones fill the last round of the persistent grid badly (q|k|v of the 8B model at 4096 tokens: 384 tiles on 256 CUs; 768 half-height tiles = three full
// rounds): 144.8 vs 130.2 us on the fp8 GEMM — a 128-row tile costs 0.74 of a 256-row one, not 0.5 — so the rule stays.
static int llm_gemm_tile_rows(int M, int tx) { return ((M + 255) / 256) * tx >= 200 ? 256 : 128; }

hipError_t llm_gemm_launch(const unsigned short *X, const void *Wp, int M, int N, int K, void *out, int ldo, int kind, hipStream_t st,
                           const float *wscale) {
    if (!X || !Wp || !out || M <= 0 || (N & 255) || (K & 127) || K < 128) return hipErrorInvalidValue;
    GemmArgs a{};
    a.X = (const f16_t *)X; a.W = (const f16_t *)Wp; a.M = M; a.N = N; a.K = K; a.ldx = K; a.ldo = ldo; a.outb = (unsigned short *)out; a.out32 = (float *)out; a.xpad = 1;
    a.bias = wscale;
    const int tx = N / 256;
    const int bm = llm_gemm_tile_rows(M, tx);
    // column groups of the 2-D XCD split: the smallest split whose W slice (N / cb columns x K) stays inside an XCD's L2 (vit_gemm.inc)
    int cb = 1;
    while (cb < 8 && tx % (cb * 2) == 0 && (size_t)(N / cb) * K * 2 > ((size_t)9 << 18)) cb *= 2;
    a.cb = cb;
    const int tiles = ((M + bm - 1) / bm) * tx, grid = std::min(tiles, vit_num_cus());
#define VLO_LLM_GO(EP_)                                                                                         \
    do {                                                                                                        \
        if (bm == 256) hipLaunchKernelGGL((vit_gemm_pp_kernel<256, EP_>), dim3(grid), dim3(512), 0, st, a);     \
        else hipLaunchKernelGGL((vit_gemm_pp_kernel<128, EP_>), dim3(grid), dim3(512), 0, st, a);               \
        return hipGetLastError();                                                                               \
    } while (0)
    if (kind == LLM_GEMM_BF16) VLO_LLM_GO(EP_LLM_BF16);
    if (kind == LLM_GEMM_SWIGLU) VLO_LLM_GO(EP_LLM_SWIGLU);
    if (kind == LLM_GEMM_RESID) VLO_LLM_GO(EP_LLM_RESID);
    if (kind == LLM_GEMM_F32) VLO_LLM_GO(EP_LLM_F32);
#undef VLO_LLM_GO
    return hipErrorInvalidValue;
}

// ---- native fp8 MFMA (prefill.h): X rows quantised to e4m3 with one fp32 scale each, W = the fp8 GEMV image as stored ------------------------
// One workgroup per row, two passes over the row (the second one hits L2): max |x|, then 16-byte chunks of codes.  Output chunk j of a row holds
// the k's of the image's register (kf2 = j / 4, quad = j % 4): 64 kf2 + 8 quad + [0, 8) and 64 kf2 + 32 + 8 quad + [0, 8) — two 16-byte loads in,
// one 16-byte store out.  scale = max|x| * (1 / 448) (1 for an all-zero row), code = e4m3_rne(clamp(x / scale, -448, 448)): the weights' rule
// (checkpoint.quantize_fp8_per_channel) applied to an activation row.
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const bf16_t *__restrict__ X, int K, uint4 *__restrict__ Xq, float *__restrict__ scale) {
    __shared__ float sm[16];
    const size_t row = blockIdx.x;
    const uint4 *src = reinterpret_cast<const uint4 *>(X + row * (size_t)K);
    const int nchunk = K >> 4;                                   // 16-byte chunks of codes = 16 k's each
    unsigned m = 0;                                              // max of the magnitudes' bit patterns (monotonic for finite bf16)
    for (int j = threadIdx.x; j < (K >> 3); j += blockDim.x) {
        const uint4 v = src[j];
        const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) m = max(m, max(w4[i] & 0x7fffu, (w4[i] >> 16) & 0x7fffu));
    }
    const float amax = block_max(__uint_as_float(m << 16), sm);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    if (threadIdx.x == 0) scale[row] = sc;
    uint4 *dst = Xq + row * (size_t)nchunk;
    auto code2 = [&](unsigned pair, unsigned old, bool hi) {     // two bf16 -> two e4m3 bytes into the low / high half of `old`
        const float a = fminf(fmaxf(__uint_as_float(pair << 16) / sc, -448.f), 448.f);
        const float b = fminf(fmaxf(__uint_as_float(pair & 0xffff0000u) / sc, -448.f), 448.f);
        return hi ? (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, true) : (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, false);
    };
    for (int j = threadIdx.x; j < nchunk; j += blockDim.x) {
        const int kf2 = j >> 2, quad = j & 3;
        const uint4 lo = src[kf2 * 8 + quad], hi = src[kf2 * 8 + 4 + quad];
        uint4 o;
        o.x = code2(lo.y, code2(lo.x, 0u, false), true);
        o.y = code2(lo.w, code2(lo.z, 0u, false), true);
        o.z = code2(hi.y, code2(hi.x, 0u, false), true);
        o.w = code2(hi.w, code2(hi.z, 0u, false), true);
        dst[j] = o;
    }
}
hipError_t quantize_rows_fp8_launch(const unsigned short *X, int M, int K, void *Xq, float *scale, hipStream_t st) {
    if (!X || !Xq || !scale || M <= 0 || K <= 0 || (K & 63)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3(M), dim3(256), 0, st, X, K, (uint4 *)Xq, scale);
    return hipGetLastError();
}

bool llm_gemm_fp8_ok(int N, int K) { return N > 0 && !(N & 255) && K >= 256 && !(K & 255); }

hipError_t llm_gemm_fp8_launch(const void *Xq, const float *xscale, const void *Wp8, const float *wscale, int M, int N, int K, void *out, int ldo,
                               int kind, hipStream_t st) {
    if (!Xq || !xscale || !Wp8 || !wscale || !out || M <= 0 || !llm_gemm_fp8_ok(N, K)) return hipErrorInvalidValue;
    GemmArgs a{};
    // the kernel's "element" is a pair of bytes (vit_gemm.inc, F8): K / 2 elements per row, K tiles of 64 elements = 128 k's
    a.X = (const f16_t *)Xq; a.W = (const f16_t *)Wp8; a.M = M; a.N = N; a.K = K / 2; a.ldx = K / 2; a.ldo = ldo;
    a.outb = (unsigned short *)out; a.out32 = (float *)out; a.xpad = 1;
    a.bias = wscale; a.rscale = xscale;
    const int tx = N / 256;
    const int bm = llm_gemm_tile_rows(M, tx);
    int cb = 1;
    while (cb < 8 && tx % (cb * 2) == 0 && (size_t)(N / cb) * K > ((size_t)9 << 18)) cb *= 2;
    a.cb = cb;
    const int tiles = ((M + bm - 1) / bm) * tx, grid = std::min(tiles, vit_num_cus());
#define VLO_LLM8_GO(EP_)                                                                                                  \
    do {                                                                                                                  \
        if (bm == 256) hipLaunchKernelGGL((vit_gemm_pp_kernel<256, EP_, 1, 0, 1>), dim3(grid), dim3(512), 0, st, a);      \
        else hipLaunchKernelGGL((vit_gemm_pp_kernel<128, EP_, 1, 0, 1>), dim3(grid), dim3(512), 0, st, a);                \
        return hipGetLastError();                                                                                         \
    } while (0)
    if (kind == LLM_GEMM_BF16) VLO_LLM8_GO(EP_LLM_BF16);
    if (kind == LLM_GEMM_SWIGLU) VLO_LLM8_GO(EP_LLM_SWIGLU);
    if (kind == LLM_GEMM_RESID) VLO_LLM8_GO(EP_LLM_RESID);
    if (kind == LLM_GEMM_F32) VLO_LLM8_GO(EP_LLM_F32);
#undef VLO_LLM8_GO
    return hipErrorInvalidValue;
}

// one thread per (token, head, 4 rotary columns): the GEMV path's EPI_ROPE on a [M][(nh + 2 nkv) hd] matrix
__global__ __launch_bounds__(256) void rope_kv_append_kernel(const bf16_t *__restrict__ qkv, int M, int nh, const bf16_t *__restrict__ cos_tab,
                                                             const bf16_t *__restrict__ sin_tab, KvGeom kv, int layer, long long pos0,
                                                             bf16_t *__restrict__ q_out) {
    const int hd = kv.head_dim, half = hd >> 1, nkv = kv.num_kv_heads, heads = nh + 2 * nkv, per_head = half >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * heads * per_head) return;
    const int i = (int)(idx % per_head) * 4, head = (int)((idx / per_head) % heads), m = (int)(idx / ((long long)per_head * heads));
    const bf16_t *src = qkv + (size_t)m * heads * hd + (size_t)head * hd;
    const ushort4 a4 = *reinterpret_cast<const ushort4 *>(src + i), b4 = *reinterpret_cast<const ushort4 *>(src + half + i);
    const bf16_t va[4] = {a4.x, a4.y, a4.z, a4.w}, vb[4] = {b4.x, b4.y, b4.z, b4.w};
    const long long pos = pos0 + m;
    const int page = kv.page_table[pos / VLO_PAGE_TOKENS], tok = (int)(pos % VLO_PAGE_TOKENS);
    if (head < nh + nkv) {
        bf16_t *dst = (head < nh) ? q_out + (size_t)m * nh * hd + (size_t)head * hd
                                  : kv.k_pool + (size_t)layer * kv.layer_stride + (size_t)page * kv.page_elems + ((size_t)(head - nh) * VLO_PAGE_TOKENS + tok) * hd;
        const ushort4 c4 = *reinterpret_cast<const ushort4 *>(cos_tab + pos * half + i), s4 = *reinterpret_cast<const ushort4 *>(sin_tab + pos * half + i);
        const bf16_t cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
        bf16_t lo[4], hi[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x1 = bf2f(va[r]), x2 = bf2f(vb[r]), c = bf2f(cc[r]), sn = bf2f(ss[r]);
            lo[r] = f2bf(rbf(x1 * c) + rbf(-x2 * sn));            // q*cos + rotate_half(q)*sin, each product and the sum rounded to bf16 (HF :157-158)
            hi[r] = f2bf(rbf(x2 * c) + rbf(x1 * sn));
        }
        *reinterpret_cast<ushort4 *>(dst + i) = *reinterpret_cast<const ushort4 *>(lo);
        *reinterpret_cast<ushort4 *>(dst + half + i) = *reinterpret_cast<const ushort4 *>(hi);
    } else {
        bf16_t *dst = kv.vt_pool + (size_t)layer * kv.layer_stride + (size_t)page * kv.page_elems + ((size_t)(head - nh - nkv) * hd) * VLO_PAGE_TOKENS + tok;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dst[(size_t)(i + r) * VLO_PAGE_TOKENS] = va[r];
            dst[(size_t)(half + i + r) * VLO_PAGE_TOKENS] = vb[r];
        }
    }
}

hipError_t rope_kv_append_launch(const unsigned short *qkv, int M, int num_heads, const unsigned short *cos_tab, const unsigned short *sin_tab,
                                 KvGeom kv, int layer, long long pos0, unsigned short *q_out, hipStream_t st) {
    if (M <= 0 || (kv.head_dim & 7)) return hipErrorInvalidValue;
    const long long total = (long long)M * (num_heads + 2 * kv.num_kv_heads) * (kv.head_dim >> 3);
    hipLaunchKernelGGL(rope_kv_append_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, qkv, M, num_heads, cos_tab, sin_tab, kv, layer,
                       pos0, q_out);
    return hipGetLastError();
}
