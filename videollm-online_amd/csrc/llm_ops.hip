// llm_ops.hip — the non-GEMV kernels of one Llama streaming step on gfx950.
//
//                           (RMSNorm, residual adds, RoPE and the KV append are fused into gemv.hip)
//   attn_chunk_kernel       n<=16 queries x growing KV, GQA, bottom-right causal mask fused,
//                           split-KV with online softmax (replaces mask build + repeat_kv + SDPA,
//                           HF:integrations/sdpa_attention.py:79-166, HF:masking_utils.py)
//   attn_combine_kernel     merge the split partials
//   embed_gather_kernel     model.get_input_embeddings() (demo/inference.py:46,66)
//   greedy / stream sample  models/modeling_live.py:177 ; demo/inference.py:76-79
//
// Rounding points mirror the reference's bf16 CPU/sdpa path (activations are bf16
// between ops, accumulation is fp32), see DESIGN.md "Numerics".
#include <stdlib.h>

#include "common.cuh"
#include "glds_asm.cuh"
#include "llm_ops.h"

// ------------------------------------------------------------------------------------
// residual add + RMSNorm.  One block per token row.
// ------------------------------------------------------------------------------------
#define RMS_THREADS 512
#define RMS_MAXCH 2   // supports H <= 8 * 512 * 2 = 8192

__global__ __launch_bounds__(RMS_THREADS) void add_rmsnorm_kernel(bf16_t *__restrict__ h, const float *__restrict__ partial,
                                                                  int ksplit, int partial_ld, const bf16_t *__restrict__ w,
                                                                  bf16_t *__restrict__ x, int H, int ldx, float eps) {
#define VLO_RMS_ROW blockIdx.x
#include "rmsnorm_body.inc"
#undef VLO_RMS_ROW
}

hipError_t add_rmsnorm_launch(unsigned short *h, const float *partial, int ksplit, int partial_ld, const unsigned short *w,
                              unsigned short *x, int H, int ldx, float eps, int n, hipStream_t st) {
    if (H > 8 * RMS_THREADS * RMS_MAXCH || (H & 7)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(add_rmsnorm_kernel, dim3(n), dim3(RMS_THREADS), 0, st, h, partial, ksplit, partial_ld, w, x, H, ldx, eps);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// chunk attention.  grid = (nsplit, nkv); block = NHG x KS waves:
//   NHG = G / HPW head groups (wave hg owns q heads kvh*G + hg*HPW .. +HPW),
//   KS  = in-block key sub-splits (wave ks walks every KS-th 32-key block of the block's chunk),
// so a CU holds 8 waves streaming different K/V pages while only one partial per (split, head)
// leaves the block: the KS partial (m, l, O) states are merged through LDS (flash-decoding inside
// the block), the block's result goes to the split-KV partial buffers, attn_combine_kernel merges
// the splits.  Per 32 keys:
//   S^T[key][qrow] = K[key][:] . Q[qrow][:]          (K page rows = MFMA A operand, from HBM)
//   online softmax per (head, qrow = lane&15); P^T stays in the lanes that produced it
//   O^T[d][qrow]  += V^T[d][key] . P^T[key][qrow]    (V^T page rows = MFMA A operand)
// ------------------------------------------------------------------------------------
template <int HD, int HPW>
__global__ __launch_bounds__(512) void attn_chunk_kernel(const bf16_t *__restrict__ q, KvGeom kv, int layer, int nh, int G, int KS,
                                                         int64_t pos0, int n, int chunk, float scale,
                                                         float *__restrict__ part_o, float *__restrict__ part_ml) {
    extern __shared__ __attribute__((aligned(16))) float4 lds_o[];        // [(KS-1)*NHG][HPW][NDT][64] float4, then m/l
#define VLO_ATTN_BX blockIdx.x
#define VLO_ATTN_BY blockIdx.y
#define VLO_ATTN_BZ blockIdx.z
#define VLO_ATTN_GX gridDim.x
#define VLO_ATTN_EXIT return
#include "attn_body.inc"
#undef VLO_ATTN_BX
#undef VLO_ATTN_BY
#undef VLO_ATTN_BZ
#undef VLO_ATTN_GX
#undef VLO_ATTN_EXIT
}

// ------------------------------------------------------------------------------------
// column-packed chunk attention (short steps: G * n <= 16 * NCT columns).  The G query heads that share one kv head TIMES the n new
// tokens are the COLUMNS of the MFMA tiles (column c = token c / G, head c % G), so one wave serves the whole GQA group and the 8
// waves of a block all walk DIFFERENT 32-key blocks: every K / V^T byte is fetched by exactly one wave of the chip (in
// attn_chunk_kernel the head-group waves of a block fetch the same pages twice, which halves the bytes in flight per CU).
//   * keys are permuted inside a 32-key block so that the lane holding S rows qd*4..+4 of both 16-key tiles owns the 8 CONSECUTIVE
//     keys qd*8..+8: P^T is a B operand as produced and a V^T fragment is one 16-byte load (two 8-byte loads in attn_chunk_kernel);
//   * Q fragments live in LDS (shared by the 8 waves), not in registers: the accumulators of 3 column tiles are 96 registers;
//   * the 8 partial states merge pairwise through LDS in three halving rounds; wave 0 writes the block's partial for the valid columns only.
// ------------------------------------------------------------------------------------
template <int HD, int NCT>
__global__ __launch_bounds__(512) void attn_cols_kernel(const bf16_t *__restrict__ q, KvGeom kv, int layer, int nh, int G, int64_t pos0, int n,
                                                        int chunk, float scale, float *__restrict__ part_o, float *__restrict__ part_ml) {
    constexpr int NKK = HD / 32, NDT = HD / 16, KS = 8;
    extern __shared__ __attribute__((aligned(16))) float4 lds_o[];         // the block's one dynamic LDS array (shared name with attn_chunk_kernel)
    uint4 *qs = reinterpret_cast<uint4 *>(lds_o);                          // [NCT][NKK][64]   Q fragments (MFMA B operand)
    float4 *lds_po = lds_o + NCT * NKK * 64;                               // [4][NCT][NDT][64] partial O of the merge rounds
    float *lds_ml = reinterpret_cast<float *>(lds_po + 4 * NCT * NDT * 64); // [4][NCT][16][2]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int split = blockIdx.x, kvh = blockIdx.y;
    const int col = lane & 15, qd = lane >> 4;
    const int L = (int)(pos0 + n);
    const int c0 = split * chunk, c1 = min(L, c0 + chunk);

    for (int i = w; i < NCT * NKK; i += KS) {
        const int ct = i / NKK, kk = i - ct * NKK;
        const int cc = ct * 16 + col, qi = cc / G, h = cc - qi * G;
        uint4 z = make_uint4(0, 0, 0, 0);
        if (qi < n) z = *reinterpret_cast<const uint4 *>(q + (size_t)qi * nh * HD + (size_t)(kvh * G + h) * HD + kk * 32 + qd * 8);
        qs[i * 64 + lane] = z;
    }
    int qpos[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) qpos[ct] = (int)pos0 + min((ct * 16 + col) / G, n - 1);

    f32x4 O[NCT][NDT];
    float mrun[NCT], lrun[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        mrun[ct] = -INFINITY;
        lrun[ct] = 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) O[ct][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bf16_t *kbase = kv.k_pool + (size_t)layer * kv.layer_stride;
    const bf16_t *vbase = kv.vt_pool + (size_t)layer * kv.layer_stride;
    const int krow = (col >> 2) * 8 + (col & 3);                           // S row `col` of tile t is key krow + 4 t of the block
    auto load_k = [&](int kt0, frag_ab (&dst)[2][NKK]) {
        const int page = kv.page_table[kt0 / VLO_PAGE_TOKENS];
        const bf16_t *kp = kbase + (size_t)page * kv.page_elems + ((size_t)kvh * VLO_PAGE_TOKENS + kt0 % VLO_PAGE_TOKENS) * HD;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk)
                dst[t][kk] = *reinterpret_cast<const frag_ab *>(kp + (size_t)(krow + 4 * t) * HD + kk * 32 + qd * 8);
    };
    frag_ab kf[2][NKK], kn[2][NKK];
    const int kfirst = c0 + w * 32;
    if (kfirst < c1) load_k(kfirst, kf);
    __syncthreads();                                                        // Q fragments staged
    for (int kt0 = kfirst; kt0 < c1; kt0 += KS * 32) {
        const int page = kv.page_table[kt0 / VLO_PAGE_TOKENS];
        const bf16_t *vp = vbase + (size_t)page * kv.page_elems + ((size_t)kvh * HD) * VLO_PAGE_TOKENS + kt0 % VLO_PAGE_TOKENS;
        frag_ab vf[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) vf[dt] = *reinterpret_cast<const frag_ab *>(vp + (size_t)(dt * 16 + col) * VLO_PAGE_TOKENS + qd * 8);
        const bool more = kt0 + KS * 32 < c1;
        if (more) load_k(kt0 + KS * 32, kn);
        asm volatile("" ::: "memory");                                     // the Q fragments are re-read from LDS every block, never hoisted into registers
        const int kb = kt0 + qd * 8;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const frag_ab qf = __builtin_bit_cast(frag_ab, qs[(ct * NKK + kk) * 64 + lane]);
                s0 = mfma_bf16(kf[0][kk], qf, s0);
                s1 = mfma_bf16(kf[1][kk], qf, s1);
            }
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = (kb + r <= qpos[ct]) ? s0[r] * scale : -INFINITY;
                v[4 + r] = (kb + 4 + r <= qpos[ct]) ? s1[r] * scale : -INFINITY;
            }
            float tmax = v[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) tmax = fmaxf(tmax, v[j]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(mrun[ct], tmax);
            const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __expf(mrun[ct] - m_safe);
            mrun[ct] = m_new;
            float psum = 0.f;
            float p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                p[j] = __expf(v[j] - m_safe);
                psum += p[j];
            }
            const uint4 pk = make_uint4(pack2bf(p[0], p[1]), pack2bf(p[2], p[3]), pack2bf(p[4], p[5]), pack2bf(p[6], p[7]));
            const frag_ab pb = __builtin_bit_cast(frag_ab, pk);
            lrun[ct] = lrun[ct] * alpha + psum;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                f32x4 o = O[ct][dt];
                o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
                O[ct][dt] = mfma_bf16(vf[dt], pb, o);
            }
        }
        if (more) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) kf[t][kk] = kn[t][kk];
        }
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        lrun[ct] += __shfl_xor(lrun[ct], 16, 64);
        lrun[ct] += __shfl_xor(lrun[ct], 32, 64);
    }
    // ---- pairwise merge of the 8 partial states: wave w + half hands its state to wave w
    for (int half = KS / 2; half >= 1; half >>= 1) {
        if (w >= half && w < 2 * half) {
            const int slot = w - half;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if (qd == 0) {
                    lds_ml[((slot * NCT + ct) * 16 + col) * 2] = mrun[ct];
                    lds_ml[((slot * NCT + ct) * 16 + col) * 2 + 1] = lrun[ct];
                }
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const f32x4 o = O[ct][dt];
                    lds_po[((size_t)(slot * NCT + ct) * NDT + dt) * 64 + lane] = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
        __syncthreads();
        if (w < half) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const float mo = lds_ml[((w * NCT + ct) * 16 + col) * 2], lo = lds_ml[((w * NCT + ct) * 16 + col) * 2 + 1];
                const float M = fmaxf(mrun[ct], mo);
                const float Ms = (M == -INFINITY) ? 0.f : M;
                const float wa = __expf(mrun[ct] - Ms), wb = __expf(mo - Ms);          // -inf -> 0
                lrun[ct] = lrun[ct] * wa + lo * wb;
                mrun[ct] = M;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const float4 o = lds_po[((size_t)(w * NCT + ct) * NDT + dt) * 64 + lane];
                    O[ct][dt][0] = O[ct][dt][0] * wa + o.x * wb;
                    O[ct][dt][1] = O[ct][dt][1] * wa + o.y * wb;
                    O[ct][dt][2] = O[ct][dt][2] * wa + o.z * wb;
                    O[ct][dt][3] = O[ct][dt][3] * wa + o.w * wb;
                }
            }
        }
        __syncthreads();
    }
    if (w != 0) return;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int cc = ct * 16 + col, qi = cc / G, h = cc - qi * G;
        if (qi >= n) continue;
        const size_t row = ((size_t)split * nh + kvh * G + h) * 16 + qi;
        if (qd == 0) {
            part_ml[row * 2] = mrun[ct];
            part_ml[row * 2 + 1] = lrun[ct];
        }
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const f32x4 o = O[ct][dt];
            *reinterpret_cast<float4 *>(part_o + row * HD + dt * 16 + qd * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// Merge of the split-KV partials.  grid = (nh, n); block = 256 threads = SL split-lanes x CL column-lanes of 4 columns (HD = 128:
// 8 x 32).  ONE memory round trip: every thread issues its float4 partial loads (<= 64 / SL of them, all independent) and the wave's
// (m, l) loads together — the split weights do not gate the partial loads — each wave works the softmax weights of the splits out
// for itself (lane s = split s: wave_max / wave_sum, no LDS), a thread picks the weights of its splits with a lane read, and one
// LDS exchange adds the SL split-lanes.  (Round 3's kernel walked the splits with two split-lanes of 4-byte loads behind a
// shared-memory weight phase: 5.9 us per layer at 32 splits, 4.8 % of the stream's GPU time, for 16 KB per block.)
template <int HD>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float *__restrict__ part_o, const float *__restrict__ part_ml,
                                                           int nsplit, int nh, bf16_t *__restrict__ out, int pack_row0) {
    constexpr int CL = HD / 4, SL = 256 / CL, NJ = VLO_MAX_SPLITS / SL;
    static_assert(VLO_MAX_SPLITS == 64, "lane s of a wave holds split s");
    __shared__ float4 red[SL * CL];
    const int head = blockIdx.x, t = threadIdx.x, lane = t & 63;
    int qrow = blockIdx.y;
    {                                              // blockIdx.y walks all query rows, 16 per sub-chunk (z = 0 on the live path)
        const int z = qrow >> 4;
        part_o += (size_t)z * nsplit * nh * 16 * HD;
        part_ml += (size_t)z * nsplit * nh * 16 * 2;
        if (pack_row0 >= 0) pack_row0 += z * 16;
        else out += (size_t)z * 16 * nh * HD;
        qrow &= 15;
    }
    const int c4 = t % CL, sl = t / CL;
    float4 o[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int sp = sl + j * SL;
        o[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sp < nsplit) o[j] = *reinterpret_cast<const float4 *>(part_o + (((size_t)sp * nh + head) * 16 + qrow) * HD + c4 * 4);
    }
    float ms = -INFINITY, ls = 0.f;
    if (lane < nsplit) {
        const float2 ml = *reinterpret_cast<const float2 *>(part_ml + (((size_t)lane * nh + head) * 16 + qrow) * 2);
        ms = ml.x;
        ls = ml.y;
    }
    const float M = wave_max(ms);
    const float wv = (ms == -INFINITY) ? 0.f : __expf(ms - M);       // 0 for lanes >= nsplit and for empty splits
    const float Ltot = wave_sum(ls * wv);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float wj = __shfl(wv, sl + j * SL, 64);
        acc.x += o[j].x * wj; acc.y += o[j].y * wj; acc.z += o[j].z * wj; acc.w += o[j].w * wj;
    }
    red[sl * CL + c4] = acc;
    __syncthreads();
    if (t < CL) {
        float4 r = red[t];
#pragma unroll
        for (int k2 = 1; k2 < SL; ++k2) {
            const float4 v = red[k2 * CL + t];
            r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
        }
        const int d = t * 4;
        const size_t at = pack_row0 < 0 ? (size_t)qrow * nh * HD + (size_t)head * HD + d : vlo_pack64_elem(pack_row0 + qrow, head * HD + d);
        ushort4 ov;
        ov.x = f2bf(r.x / Ltot); ov.y = f2bf(r.y / Ltot); ov.z = f2bf(r.z / Ltot); ov.w = f2bf(r.w / Ltot);
        *reinterpret_cast<ushort4 *>(out + at) = ov;
    }
}
static hipError_t attn_combine_launch(const float *part_o, const float *part_ml, int nsplit, int nh, int hd, int n, unsigned short *out,
                                      int pack_row0, hipStream_t st) {
    if (hd == 128) hipLaunchKernelGGL(attn_combine_kernel<128>, dim3(nh, n), dim3(256), 0, st, part_o, part_ml, nsplit, nh, out, pack_row0);
    else if (hd == 64) hipLaunchKernelGGL(attn_combine_kernel<64>, dim3(nh, n), dim3(256), 0, st, part_o, part_ml, nsplit, nh, out, pack_row0);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// max over lanes l, l ^ 16, l ^ 32, l ^ 48 (the four lanes that hold one score column of a 16 x 16 MFMA tile) by two row swaps
VLO_DEV float quad_lanes_maxf(float x) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// ------------------------------------------------------------------------------------
// Prefill attention (blocks of hundreds to thousands of new tokens, engine.hip::run_prefill): flash-style.  One workgroup = one kv head x
// QB = 128 / G consecutive queries: its 128 (query, head) columns are 8 column tiles, one per wave (8 waves), and ALL waves walk the SAME
// 32-key tiles, which are staged once per workgroup in LDS (K 32 x HD, V^T HD x 32: 16 KiB at HD = 128, double-buffered) by direct-to-LDS
// loads — a K / V^T byte leaves L2 once per 128 columns instead of once per 16-query sub-chunk (the decode-shaped kernel above re-reads
// the whole prefix for every sub-chunk: 232 TFLOP/s at 13 k tokens).  No split-KV, no merge kernel: a workgroup sees every key its
// queries may attend to and writes normalised bf16 rows.
//   * LDS layout = the MFMA A-operand fragments themselves: piece (1 KiB) = [16-byte chunk c][row r] so lane l = 16 c + r reads slot l
//     (conflict-free ds_read_b128); the direct-to-LDS destination is lane-linear, so the gather happens on the per-lane SOURCE address;
//   * keys are permuted inside a 32-key tile as in attn_cols_kernel (row r of key tile t = key (r >> 2) * 8 + (r & 3) + 4 t): the lane
//     that holds S rows 4 qd .. 4 qd + 3 of both tiles owns 8 CONSECUTIVE keys, P^T is a B operand as produced and a V^T fragment is one chunk;
//   * the loads are issued in inline asm (invisible to hipcc, which would drain them before every LDS read) into a ring of NS tile buffers and
//     retired by ONE COUNTED s_waitcnt vmcnt + a raw s_barrier per tile: NS - 1 tiles are in flight while one is multiplied (one 16-KiB tile per CU
//     in flight — NS = 2 — leaves the kernel waiting for L2 latency: a tile's MFMAs take ~0.5 us, its round trip ~2).
//   * two wave-uniform shortcuts drop work whose result is known, BIT-IDENTICALLY: a tile every key of which is visible to every query of the
//     wave skips the causal select; a tile that raised no lane's running maximum skips the rescale of the 32 accumulator registers (alpha == 1
//     exactly).  `noskip` (VLO_ATTN_NOSKIP=1, tests) takes the long way everywhere.  Measured: ~40 % fewer VALU instructions per tile and NO
//     change in time (13 312 tokens: 255.3 vs 255.8 ms) — VALU throughput is not what bounds this kernel.
// grid = (ceil(n / QB), nkv); 512 threads.  Rounding points as the other attention kernels (P -> bf16 before P.V, bf16 output).
// ------------------------------------------------------------------------------------
// ONE column tile per wave: 128 (query, head) columns per workgroup, a register budget of 128 — four waves per SIMD.  (Round 4 shipped two column tiles
// per wave, 256 columns per workgroup at 208 - 214 VGPRs / two waves per SIMD: half the LDS read traffic per FLOP, but a per-wave latency chain that two
// waves per SIMD cannot hide.  Measured on the MI355X, bit-identical outputs: 13 312 tokens 242.2 -> 232.4 ms, 2 048 tokens 33.7 -> 33.9 ms;
// profiles/r5_prefill_attention_one_column_tile.txt.  The two-tile kernel is gone.)
template <int HD, int G, int NS>
__global__ __launch_bounds__(512, 4) void attn_prefill_kernel(const bf16_t *__restrict__ q, KvGeom kv, int layer, int nh, int64_t pos0, int n, float scale_l2e,
                                                              bf16_t *__restrict__ out, int noskip, int nkv, int nqb) {
#define VLO_PF_NCT 1
#include "attn_prefill_body.inc"
#undef VLO_PF_NCT
}

// The same tiles as a two-group ping-pong with batched fragment reads (attn_prefill_pp_body.inc) — the kernel that ships; the lock-step kernel
// above stays as the reference the variants test compares it with, bit for bit (VLO_ATTN_PF=0).
template <int HD, int G, int NS>
__global__ __launch_bounds__(512, 4) void attn_prefill_pp_kernel(const bf16_t *__restrict__ q, KvGeom kv, int layer, int nh, int64_t pos0, int n, float scale_l2e,
                                                                 bf16_t *__restrict__ out, int noskip, int nkv, int nqb) {
#define VLO_PF_NCT 1
#include "attn_prefill_pp_body.inc"
#undef VLO_PF_NCT
}
// the (head dim, GQA group) pairs attn_prefill_kernel is instantiated for — attention_prefill_launch returns hipErrorNotSupported for any other; a caller
// WITHOUT a fallback (tp.hip::tp_prefill) asks first and keeps the 16-row step instead
bool attention_prefill_supported(int head_dim, int gqa_group) {
    return (head_dim == 128 && (gqa_group == 1 || gqa_group == 2 || gqa_group == 4 || gqa_group == 8)) ||
           (head_dim == 64 && (gqa_group == 2 || gqa_group == 4 || gqa_group == 8));
}

hipError_t attention_prefill_launch(const unsigned short *q, KvGeom kv, int layer, int num_heads, int64_t pos0, int n, unsigned short *out, hipStream_t st) {
    const int nkv = kv.num_kv_heads, hd = kv.head_dim, G = num_heads / nkv;
    if (n <= 0 || nkv * G != num_heads) return hipErrorInvalidValue;
    const float scale = (1.0f / sqrtf((float)hd)) * 1.4426950408889634f;   // log2(e) / sqrt(hd): the kernels' softmax runs on exp2 (attn_prefill_pp_body.inc)
    const char *ns = getenv("VLO_ATTN_NOSKIP");                            // read per call: the tests flip it between two passes over the same input
    const int noskip = ns && atoi(ns) != 0;
    const char *pv = getenv("VLO_ATTN_PF");                                // tests: 0 = the lock-step reference kernel
    const int variant = pv ? atoi(pv) : 1;
#define VLO_ATTN_PF(HD_, G_)                                                                                                          \
    do {                                                                                                                              \
        constexpr int QB_ = 128 / G_;                                                                                                 \
        const int nqb = (n + QB_ - 1) / QB_;                                                                                          \
        const dim3 grid((unsigned)((nqb * nkv + 7) & ~7));               /* 1-D: the kernel maps workgroups to (kv head, query block) per XCD */ \
        if (variant != 0)                                                                                                             \
            hipLaunchKernelGGL((attn_prefill_pp_kernel<HD_, G_, 4>), grid, dim3(512), 0, st, q, kv, layer, num_heads, pos0, n, scale, out, noskip, nkv, nqb); \
        else                                                                                                                          \
            hipLaunchKernelGGL((attn_prefill_kernel<HD_, G_, 4>), grid, dim3(512), 0, st, q, kv, layer, num_heads, pos0, n, scale, out, noskip, nkv, nqb); \
        return hipGetLastError();                                                                                                     \
    } while (0)
    if (hd == 128 && G == 4) VLO_ATTN_PF(128, 4);
    if (hd == 128 && G == 8) VLO_ATTN_PF(128, 8);
    if (hd == 128 && G == 2) VLO_ATTN_PF(128, 2);
    if (hd == 128 && G == 1) VLO_ATTN_PF(128, 1);
    if (hd == 64 && G == 8) VLO_ATTN_PF(64, 8);
    if (hd == 64 && G == 4) VLO_ATTN_PF(64, 4);
    if (hd == 64 && G == 2) VLO_ATTN_PF(64, 2);
#undef VLO_ATTN_PF
    return hipErrorNotSupported;                 // the caller falls back to attention_launch
}

hipError_t attention_geometry(const KvGeom &kv, int num_heads, int64_t pos0, int n, AttnGeom *g, int part_cap) {
    const int nkv = kv.num_kv_heads, hd = kv.head_dim, G = num_heads / nkv;
    const int L = (int)(pos0 + n);
    const int hpw = (G % 2 == 0) ? 2 : 1;
    const int nhg = G / hpw;
    if (nhg > 8) return hipErrorInvalidValue;
    int KS = 8 / nhg;                                   // 8 waves per block
    if (KS > 4) KS = 4;
    // n > 16 (block path): grid.z sub-chunks of 16 queries share one launch and one split geometry; sub-chunk z sees the
    // keys [0, pos0 + 16 z + n_z), splits beyond that write empty partials
    const int nz = (n + 15) / 16;
    if (nz > part_cap || nz > 65535) return hipErrorInvalidValue;
    // short steps whose G * n (head, token) columns fit 3 MFMA column tiles take the column-packed kernel: 8 key sub-splits per block
    g->nct = 0;
    if (nz == 1 && G * n <= 48 && (hd == 128 || hd == 64)) {
        g->nct = (G * n + 15) / 16;
        KS = 8;
    }
    // splits: ~one block per CU at long context; every wave should see at least one 32-key block
    int target = (L + KS * 32 - 1) / (KS * 32);
    constexpr int want_blocks = 256;                    // (128 / 512 measured slower)
    const int want = (want_blocks + nkv * nz - 1) / (nkv * nz);
    if (target > want) target = want;
    if (target > VLO_MAX_SPLITS / nz) target = VLO_MAX_SPLITS / nz;      // (the merge kernel holds one split per lane; 0 for nz > 64: one split below)
    if (target > part_cap / nz) target = part_cap / nz;
    if (nz > 4) target = 1;          // prefill blocks: the sub-chunks alone fill the chip, and one split keeps a row's result independent of the block's length
    if (target < 1) target = 1;
    int chunk = (L + target - 1) / target;
    chunk = (chunk + 31) & ~31;
    g->G = G; g->KS = KS; g->hpw = hpw; g->nhg = nhg; g->nz = nz; g->chunk = chunk;
    g->nsplit = (L + chunk - 1) / chunk;
    g->scale = 1.0f / sqrtf((float)hd);
    g->lds_bytes = (size_t)(KS - 1) * nhg * hpw * ((size_t)(hd / 16) * 64 * 16 + 16 * 2 * 4);
    if (g->nct) g->lds_bytes = (size_t)g->nct * ((size_t)(hd / 32) * 64 * 16 + 4 * ((size_t)(hd / 16) * 64 * 16 + 16 * 2 * 4));
    return hipSuccess;
}

hipError_t attention_launch(const unsigned short *q, KvGeom kv, int layer, int num_heads, int64_t pos0, int n,
                            float *part_o, float *part_ml, unsigned short *out, hipStream_t st, int pack_row0, int part_cap) {
    AttnGeom ag;
    const hipError_t ge = attention_geometry(kv, num_heads, pos0, n, &ag, part_cap);
    if (ge != hipSuccess) return ge;
    const int nkv = kv.num_kv_heads, hd = kv.head_dim, G = ag.G, hpw = ag.hpw, nhg = ag.nhg, KS = ag.KS, nz = ag.nz, chunk = ag.chunk,
              nsplit = ag.nsplit;
    const float scale = ag.scale;
    dim3 grid(nsplit, nkv, nz), block(nhg * KS * 64);
    const size_t lds = ag.lds_bytes;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)attn_chunk_kernel<128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)attn_chunk_kernel<128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)attn_chunk_kernel<64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)attn_chunk_kernel<64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)attn_cols_kernel<128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)attn_cols_kernel<128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)attn_cols_kernel<128, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)attn_cols_kernel<64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)attn_cols_kernel<64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)attn_cols_kernel<64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        attr_done = true;
    }
#define VLO_ATTN_COLS(HD_, NCT_) \
    hipLaunchKernelGGL((attn_cols_kernel<HD_, NCT_>), grid, dim3(512), lds, st, q, kv, layer, num_heads, G, pos0, n, chunk, scale, part_o, part_ml)
    if (ag.nct) {
        if (hd == 128 && ag.nct == 1) VLO_ATTN_COLS(128, 1);
        else if (hd == 128 && ag.nct == 2) VLO_ATTN_COLS(128, 2);
        else if (hd == 128) VLO_ATTN_COLS(128, 3);
        else if (ag.nct == 1) VLO_ATTN_COLS(64, 1);
        else if (ag.nct == 2) VLO_ATTN_COLS(64, 2);
        else VLO_ATTN_COLS(64, 3);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        return attn_combine_launch(part_o, part_ml, nsplit, num_heads, hd, n, out, pack_row0, st);
    }
#undef VLO_ATTN_COLS
#define VLO_ATTN(HD_, HPW_) \
    hipLaunchKernelGGL((attn_chunk_kernel<HD_, HPW_>), grid, block, lds, st, q, kv, layer, num_heads, G, KS, pos0, n, chunk, scale, part_o, part_ml)
    if (hd == 128 && hpw == 2) VLO_ATTN(128, 2);
    else if (hd == 128 && hpw == 1) VLO_ATTN(128, 1);
    else if (hd == 64 && hpw == 2) VLO_ATTN(64, 2);
    else if (hd == 64 && hpw == 1) VLO_ATTN(64, 1);
    else return hipErrorInvalidValue;
#undef VLO_ATTN
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return attn_combine_launch(part_o, part_ml, nsplit, num_heads, hd, n, out, pack_row0, st);
}

// ------------------------------------------------------------------------------------
// embedding gather, row copy, KV read-back (tests)
// ------------------------------------------------------------------------------------
__global__ void embed_gather_kernel(const bf16_t *__restrict__ table, const int64_t *__restrict__ ids, int H, int64_t vocab,
                                    bf16_t *__restrict__ out) {
    int64_t id = ids[blockIdx.x];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const uint4 *src = reinterpret_cast<const uint4 *>(table + (size_t)id * H);
    uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)blockIdx.x * H);
    for (int i = threadIdx.x; i < H / 8; i += blockDim.x) dst[i] = src[i];
}
hipError_t embed_gather_launch(const unsigned short *table, const int64_t *ids, int k, int H, int64_t vocab,
                               unsigned short *out, hipStream_t st) {
    hipLaunchKernelGGL(embed_gather_kernel, dim3(k), dim3(256), 0, st, table, ids, H, vocab, out);
    return hipGetLastError();
}

// step input (demo/inference.py:65-68: `torch.cat([embed(last_ids), frame_embeds])`) in ONE launch: the token ids arrive from the HOST
// as kernel arguments (no `torch.tensor(ids, device=...)` upload), blocks [0, k) gather their embedding row, blocks [k, k + rows) copy
// a frame-token row of the connector's output.
__global__ void step_input_kernel(const bf16_t *__restrict__ table, StepIds ids, int k, const bf16_t *__restrict__ frame_rows, int H,
                                  int64_t vocab, bf16_t *__restrict__ out) {
    const int r = blockIdx.x;
    const uint4 *src;
    if (r < k) {
        int64_t id = ids.v[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        src = reinterpret_cast<const uint4 *>(table + (size_t)id * H);
    } else {
        src = reinterpret_cast<const uint4 *>(frame_rows + (size_t)(r - k) * H);
    }
    uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)r * H);
    for (int i = threadIdx.x; i < H / 8; i += blockDim.x) dst[i] = src[i];
}
hipError_t step_input_launch(const unsigned short *table, const StepIds &ids, int k, const unsigned short *frame_rows, int rows, int H,
                             int64_t vocab, unsigned short *out, hipStream_t st) {
    hipLaunchKernelGGL(step_input_kernel, dim3(k + rows), dim3(256), 0, st, table, ids, k, frame_rows, H, vocab, out);
    return hipGetLastError();
}

__global__ void copy_rows_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
hipError_t copy_rows_launch(const unsigned short *src, unsigned short *dst, int rows, int H, hipStream_t st) {
    const size_t n16 = (size_t)rows * H / 8;
    int blocks = (int)((n16 + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(blocks), dim3(256), 0, st, (const uint4 *)src, (uint4 *)dst, n16);
    return hipGetLastError();
}

__global__ void read_kv_kernel(KvGeom kv, int layer, int which, int kvh, int64_t t0, bf16_t *__restrict__ dst) {
    const int64_t t = t0 + blockIdx.x;
    const int page = kv.page_table[t / VLO_PAGE_TOKENS];
    const int tok = (int)(t % VLO_PAGE_TOKENS);
    const int hd = kv.head_dim;
    for (int d = threadIdx.x; d < hd; d += blockDim.x) {
        bf16_t v;
        if (which == 0)
            v = kv.k_pool[(size_t)layer * kv.layer_stride + (size_t)page * kv.page_elems + ((size_t)kvh * VLO_PAGE_TOKENS + tok) * hd + d];
        else
            v = kv.vt_pool[(size_t)layer * kv.layer_stride + (size_t)page * kv.page_elems + ((size_t)kvh * hd + d) * VLO_PAGE_TOKENS + tok];
        dst[(size_t)blockIdx.x * hd + d] = v;
    }
}
hipError_t read_kv_launch(KvGeom kv, int layer, int which, int kv_head, int64_t t0, int64_t t1, unsigned short *dst,
                          hipStream_t st) {
    if (t1 <= t0) return hipSuccess;
    hipLaunchKernelGGL(read_kv_kernel, dim3((unsigned)(t1 - t0)), dim3(64), 0, st, kv, layer, which, kv_head, t0, dst);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// samplers over the L2-resident bf16 logits [V]: SAMPLE_BLOCKS blocks scan a slice each, a one-wave
// kernel merges the partials (3 short launches instead of one ~90 us single-block scan).
// ------------------------------------------------------------------------------------
#define SAMPLE_BLOCKS 64
#define SAMPLE_THREADS 256

struct ArgBest { float v; int i; };
VLO_DEV ArgBest better(ArgBest a, ArgBest b) {      // larger value wins; ties -> smaller index (torch argmax)
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}
VLO_DEV ArgBest wave_argbest(ArgBest x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgBest y;
        y.v = __shfl_xor(x.v, o, 64);
        y.i = __shfl_xor(x.i, o, 64);
        x = better(x, y);
    }
    return x;
}
VLO_DEV ArgBest block_argbest(ArgBest x, float *smv, int *smi) {
    x = wave_argbest(x);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) { smv[w] = x.v; smi[w] = x.i; }
    __syncthreads();
    ArgBest r = {smv[0], smi[0]};
    for (int k = 1; k < nw; ++k) r = better(r, (ArgBest){smv[k], smi[k]});
    return r;
}

// scratch layout (floats): [0, NB) block max | [NB, 2NB) block sum of exp(x - block max) | [2NB, 3NB) best value |
//                          [3NB, 4NB) best index (int bits)
__global__ __launch_bounds__(SAMPLE_THREADS) void sample_stats_kernel(const bf16_t *__restrict__ logits, int V, float *__restrict__ scr) {
    __shared__ float sm[16];
    __shared__ float smv[16];
    __shared__ int smi[16];
    const int per = (V + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = min(V, lo + per);
    float mx = -INFINITY;
    ArgBest b = {-INFINITY, 0x7fffffff};
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const float v = bf2f(logits[i]);
        mx = fmaxf(mx, v);
        if (v > b.v) { b.v = v; b.i = i; }
    }
    mx = block_max(mx, sm);
    float s = 0.f;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) s += expf(bf2f(logits[i]) - mx);
    s = block_sum(s, sm);
    b = block_argbest(b, smv, smi);
    if (threadIdx.x == 0) {
        const int NB = gridDim.x;
        scr[blockIdx.x] = mx;
        scr[NB + blockIdx.x] = (mx == -INFINITY) ? 0.f : s;
        scr[2 * NB + blockIdx.x] = b.v;
        reinterpret_cast<int *>(scr)[3 * NB + blockIdx.x] = b.i;
    }
}

// force_mode: 0 = plain argmax; 1 = argmax but never eos (scheduled mode, mid-response);
//             2 = argmax computed, eos written (scheduled mode, last token)
__global__ __launch_bounds__(64) void greedy_final_kernel(const float *__restrict__ scr, int NB, int V, int64_t *tok_out, int eos,
                                                          int force_mode) {
    ArgBest b = {-INFINITY, 0x7fffffff};
    for (int k = threadIdx.x; k < NB; k += 64) b = better(b, (ArgBest){scr[2 * NB + k], reinterpret_cast<const int *>(scr)[3 * NB + k]});
    b = wave_argbest(b);
    if (threadIdx.x == 0) {
        int t = b.i;
        if (force_mode == 1 && t == eos) t = (eos + 1) % V;
        if (force_mode == 2) t = eos;
        *tok_out = t;
    }
}
hipError_t greedy_sample_launch(const unsigned short *logits, int V, int64_t *tok_out, int eos, int force_mode, float *scratch,
                                hipStream_t st) {
    hipLaunchKernelGGL(sample_stats_kernel, dim3(SAMPLE_BLOCKS), dim3(SAMPLE_THREADS), 0, st, logits, V, scratch);
    hipLaunchKernelGGL(greedy_final_kernel, dim3(1), dim3(64), 0, st, scratch, SAMPLE_BLOCKS, V, tok_out, eos, force_mode);
    return hipGetLastError();
}

// demo/inference.py:76-79: softmax over a bf16 tensor (fp32 inside, bf16 out), threshold on p[interval], argmax of p.
// Every block recomputes the global (max, sum) from the NB partials, then scans its slice of p.
__global__ __launch_bounds__(SAMPLE_THREADS) void stream_scan_kernel(const bf16_t *__restrict__ logits, int V, float threshold,
                                                                     int interval_id, float *__restrict__ scr) {
    __shared__ float smv[16];
    __shared__ int smi[16];
    const int NB = gridDim.x;
    float M = -INFINITY;
    for (int k = 0; k < NB; ++k) M = fmaxf(M, scr[k]);
    float S = 0.f;
    for (int k = 0; k < NB; ++k) S += scr[NB + k] * expf(scr[k] - M);
    const float p_int = rbf(expf(bf2f(logits[interval_id]) - M) / S);
    const bool zero_int = p_int < rbf(threshold);   // torch compares a bf16 tensor with a Python float in bf16
    const int per = (V + NB - 1) / NB, lo = blockIdx.x * per, hi = min(V, lo + per);
    ArgBest b = {-INFINITY, 0x7fffffff};
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        float p = rbf(expf(bf2f(logits[i]) - M) / S);
        if (i == interval_id && zero_int) p = 0.f;
        if (p > b.v) { b.v = p; b.i = i; }
    }
    b = block_argbest(b, smv, smi);
    if (threadIdx.x == 0) {
        scr[4 * NB + blockIdx.x] = b.v;
        reinterpret_cast<int *>(scr)[5 * NB + blockIdx.x] = b.i;
        if (blockIdx.x == 0) scr[6 * NB] = p_int;
    }
}
__global__ __launch_bounds__(64) void stream_final_kernel(const float *__restrict__ scr, int NB, int64_t *tok_out, float *p_interval_out) {
    ArgBest b = {-INFINITY, 0x7fffffff};
    for (int k = threadIdx.x; k < NB; k += 64) b = better(b, (ArgBest){scr[4 * NB + k], reinterpret_cast<const int *>(scr)[5 * NB + k]});
    b = wave_argbest(b);
    if (threadIdx.x == 0) {
        *tok_out = b.i;
        if (p_interval_out) *p_interval_out = scr[6 * NB];
    }
}
hipError_t stream_sample_launch(const unsigned short *logits, int V, float threshold, int interval_id, int64_t *tok_out,
                                float *p_interval_out, float *scratch, hipStream_t st) {
    if (interval_id < 0 || interval_id >= V) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sample_stats_kernel, dim3(SAMPLE_BLOCKS), dim3(SAMPLE_THREADS), 0, st, logits, V, scratch);
    hipLaunchKernelGGL(stream_scan_kernel, dim3(SAMPLE_BLOCKS), dim3(SAMPLE_THREADS), 0, st, logits, V, threshold, interval_id, scratch);
    hipLaunchKernelGGL(stream_final_kernel, dim3(1), dim3(64), 0, st, scratch, SAMPLE_BLOCKS, tok_out, p_interval_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Teacher-forced evaluation helpers (models/modeling_live.py:29-42, 44-168, 170-171)
// ------------------------------------------------------------------------------------
// joint_embed (:38-41): rank of every placeholder position among the placeholders (exclusive scan, one block).
__global__ __launch_bounds__(1024) void placeholder_rank_kernel(const int64_t *__restrict__ ids, int k, int64_t v_id,
                                                                int *__restrict__ src_idx, int *__restrict__ count_out) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < k; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const int f = (i < k && ids[i] == v_id) ? 1 : 0;
        int inc = f;                                   // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(inc, o, 64);
            if (lane >= o) inc += y;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int before = base_s;
        for (int j = 0; j < w; ++j) before += wsum[j];
        if (i < k) src_idx[i] = f ? before + inc - 1 : -1;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int j = 0; j < 16; ++j) t += wsum[j];
            base_s += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count_out = base_s;
}
// rows with src_idx >= 0 take frame-token row src_idx, the others the (clamped) embedding-table row
__global__ void joint_gather_kernel(const bf16_t *__restrict__ table, const int64_t *__restrict__ ids, const int *__restrict__ src_idx,
                                    const bf16_t *__restrict__ frame_rows, int n_frame_rows, int H, int64_t vocab,
                                    bf16_t *__restrict__ out) {
    const int r = src_idx[blockIdx.x];
    const uint4 *src;
    if (r >= 0) {
        if (r >= n_frame_rows) return;                 // count mismatch: the host reports the error
        src = reinterpret_cast<const uint4 *>(frame_rows + (size_t)r * H);
    } else {
        int64_t id = ids[blockIdx.x];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        src = reinterpret_cast<const uint4 *>(table + (size_t)id * H);
    }
    uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)blockIdx.x * H);
    for (int i = threadIdx.x; i < H / 8; i += blockDim.x) dst[i] = src[i];
}
hipError_t joint_embed_launch(const unsigned short *table, const int64_t *ids, int k, int64_t v_id, const unsigned short *frame_rows,
                              int n_frame_rows, int H, int64_t vocab, int *src_idx_scratch, int *count_out, unsigned short *out,
                              hipStream_t st) {
    hipLaunchKernelGGL(placeholder_rank_kernel, dim3(1), dim3(1024), 0, st, ids, k, v_id, src_idx_scratch, count_out);
    hipLaunchKernelGGL(joint_gather_kernel, dim3(k), dim3(256), 0, st, table, ids, src_idx_scratch, frame_rows, n_frame_rows, H, vocab,
                       out);
    return hipGetLastError();
}

// trim_past_key_values(past, 0, stop) as a fork: copy the pages holding positions [0, stop) of every layer into the
// pages of another session.  grid = (pages, layers, 2 {K, V^T}).
__global__ void kv_copy_pages_kernel(KvGeom kv, const int *__restrict__ src_pt, const int *__restrict__ dst_pt) {
    unsigned short *pool = blockIdx.z == 0 ? kv.k_pool : kv.vt_pool;
    const size_t lay = (size_t)blockIdx.y * kv.layer_stride;
    const uint4 *src = reinterpret_cast<const uint4 *>(pool + lay + (size_t)src_pt[blockIdx.x] * kv.page_elems);
    uint4 *dst = reinterpret_cast<uint4 *>(pool + lay + (size_t)dst_pt[blockIdx.x] * kv.page_elems);
    const int n16 = (int)(kv.page_elems / 8);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
}
hipError_t kv_copy_pages_launch(KvGeom kv, const int *src_pt, const int *dst_pt, int pages, int layers, hipStream_t st) {
    if (pages <= 0) return hipSuccess;
    hipLaunchKernelGGL(kv_copy_pages_kernel, dim3(pages, layers, 2), dim3(256), 0, st, kv, src_pt, dst_pt);
    return hipGetLastError();
}

// Per-row statistics of a bf16 logits matrix [n][ld] (one block per row): everything stream_evaluate reads from the
// logits (models/modeling_live.py:95-97, 107-112, 140-144) without materialising softmax rows:
//   lse[r]          log sum exp (fp32)                  -> cross entropy = lse - label_logit
//   argmax[r]       first maximum of the logits         (:97)
//   label_logit[r]  logits[r][labels[r]] (0 when the label is outside [0, V))
//   p_interval[r]   softmax(logits)[interval] rounded to bf16 as the reference's bf16 softmax does (:107,:110)
//   p_argmax[r]     first maximum of the bf16-rounded softmax row (:112; rounding can merge near-ties into exact ties,
//                   which argmax then resolves to the lower index — not always the logits' argmax)
__global__ __launch_bounds__(256) void logit_rows_kernel(const bf16_t *__restrict__ logits, int V, int64_t ld,
                                                         const int64_t *__restrict__ labels, int interval_id, float *__restrict__ lse,
                                                         int64_t *__restrict__ amax, float *__restrict__ label_logit,
                                                         float *__restrict__ p_interval, int64_t *__restrict__ p_amax) {
    __shared__ float sm[16];
    __shared__ float smv[16];
    __shared__ int smi[16];
    const bf16_t *x = logits + (size_t)blockIdx.x * ld;
    float mx = -INFINITY;
    ArgBest b = {-INFINITY, 0x7fffffff};
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        const float v = bf2f(x[i]);
        mx = fmaxf(mx, v);
        if (v > b.v) { b.v = v; b.i = i; }
    }
    const float M = block_max(mx, sm);
    float s = 0.f;
    for (int i = threadIdx.x; i < V; i += blockDim.x) s += expf(bf2f(x[i]) - M);
    const float S = block_sum(s, sm);
    b = block_argbest(b, smv, smi);
    ArgBest pb = {-INFINITY, 0x7fffffff};
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        const float p = rbf(expf(bf2f(x[i]) - M) / S);
        if (p > pb.v) { pb.v = p; pb.i = i; }
    }
    pb = block_argbest(pb, smv, smi);
    if (threadIdx.x == 0) {
        const int r = blockIdx.x;
        lse[r] = M + logf(S);
        amax[r] = b.i;
        const int64_t lab = labels ? labels[r] : -1;
        label_logit[r] = (lab >= 0 && lab < V) ? bf2f(x[lab]) : 0.f;
        p_interval[r] = (interval_id >= 0 && interval_id < V) ? rbf(expf(bf2f(x[interval_id]) - M) / S) : 0.f;
        p_amax[r] = pb.i;
    }
}
hipError_t logit_rows_launch(const unsigned short *logits, int n, int V, int64_t ld, const int64_t *labels, int interval_id, float *lse,
                             int64_t *amax, float *label_logit, float *p_interval, int64_t *p_amax, hipStream_t st) {
    hipLaunchKernelGGL(logit_rows_kernel, dim3(n), dim3(256), 0, st, logits, V, ld, labels, interval_id, lse, amax, label_logit,
                       p_interval, p_amax);
    return hipGetLastError();
}
