// gemv.hip — weight-streaming skinny GEMM for the Llama step (n <= 16 rows), with the neighbouring
// elementwise work fused into its operand load and epilogue.
//
// y[m][n] = sum_k x[m][k] * W[n][k]      x: bf16 [16][ldx] row-major (rows >= n_rows ignored)
//                                        W: bf16, HF nn.Linear layout [N][K]
//
// Replaces the hipBLASLt/cuBLAS skinny GEMMs under q/k/v/o_proj, gate/up/down_proj, lm_head
// (HF:models/llama/modeling_llama.py:174-176,254-256,280,477-480) and the connector Linears
// (models/live_llama/modeling_live_llama.py:18-22), plus — fused —
//   XSRC_NORM   LlamaRMSNorm of the residual stream while building the activation fragments (:62-67)
//   EPI_RESID   `hidden_states = residual + hidden_states` (:317, :323) and the next norm's row sum of squares
//   EPI_ROPE    q/k/v split, apply_rotary_pos_emb (:138-160), KV append (replaces DynamicLayer.update's
//               torch.cat of the whole cache, HF:cache_utils.py:127-151)
//   EPI_SWIGLU  act_fn(gate) * up (:176);  EPI_BF16_GELU_ERF  the connector's python-GELU
// so a decoder layer is 6 launches (qkv, attention, combine, o, gate_up, down) instead of 9+.
//
// HBM-bound (arithmetic intensity ~ n FLOP per weight byte): every weight byte is streamed exactly
// once with 1-KiB-per-wave coalesced global_load_dwordx4; the tiny activation operand lives in VGPRs.
//
// Weight image in HBM ("packed", built once at load time by pack_weight_kernel):
//   Wp[tile][kf][lane] : 16 bytes = W[tile*16 + (lane&15)][kf*32 + (lane>>4)*8 .. +8]
// = the A-operand fragment of v_mfma_f32_16x16x32_bf16.  The MFMA computes
//   D[nrow][m] = sum_k Wfrag[nrow][k] * xfrag[k][m]   (W as A, x as B):
//   lane l, reg r  ->  output column n = tile*16 + (l>>4)*4 + r, token row m = l&15.
//
// Work split: persistent blocks (grid.x ~ one per CU) stride over GROUPS of two column tiles;
// grid.y = K slices (unit tests only; the step uses whole-K kernels so epilogues see final sums).
// Inside a block the NW waves split K; a wave walks its K range in KC chunks of KF fragments and
// keeps the chunk's activation fragments in VGPRs.  Weight fragment kf is re-loaded for the next
// (chunk, tile) item right after the MFMA that consumed it (rolling prefetch): KF KiB of HBM reads
// stay in flight per wave with a single register set.  Per group: partial tiles -> LDS
// (double-buffered, one barrier), the first waves reduce across waves and run the epilogue.
#include <stdlib.h>

#include "common.cuh"
#include "gemv.h"

// ------------------------------------------------------------------------------------
// packing: row-major [N][K] bf16 -> fragment order.  Rows >= N_valid are zero-filled.
// Source tile t lands at destination tile t*tile_stride + tile_offset (used to concatenate q/k/v and
// to interleave gate/up 16-row tiles for the SwiGLU epilogue).
// ------------------------------------------------------------------------------------
// `ldw` = source row stride in elements (>= K): lets a tensor-parallel rank pack a column slice of W.
// `half` = -1: whole 16-row tiles.  half = 0 / 1: 8-row interleave — source rows 8t..8t+7 land in rows 0..7 (half 0) or
// 8..15 (half 1) of destination tile t and the other 8 lanes-rows are left untouched (gate/up share a tile so that the
// SwiGLU epilogue finds both in ONE tile and I/8 single tiles balance over the 256 CUs).
__global__ void pack_weight_kernel(const bf16_t *__restrict__ W, uint4 *__restrict__ Wp, int N_valid, int K, int ldw,
                                   int NT, int KFtot, int tile_stride, int tile_offset, int half) {
    const size_t total = (size_t)NT * KFtot * 64;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t t = i >> 6;
        const int kf = (int)(t % KFtot);
        const int tile = (int)(t / KFtot);
        const int k = kf * 32 + (lane >> 4) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (half < 0) {
            const int n = tile * 16 + (lane & 15);
            if (n < N_valid) v = *reinterpret_cast<const uint4 *>(W + (size_t)n * ldw + k);
        } else {
            const int r16 = lane & 15;
            if ((r16 >> 3) != half) continue;                     // the other half belongs to the other matrix
            const int n = tile * 8 + (r16 & 7);
            if (n < N_valid) v = *reinterpret_cast<const uint4 *>(W + (size_t)n * ldw + k);
        }
        Wp[((size_t)(tile * tile_stride + tile_offset) * KFtot + kf) * 64 + lane] = v;
    }
}

// ------------------------------------------------------------------------------------
// epilogue math (rounding points follow the reference's bf16 CPU path)
// ------------------------------------------------------------------------------------
VLO_DEV float silu_bf16(float g) {           // F.silu on a bf16 tensor: fp32 math, one rounding
    return rbf(g / (1.0f + __expf(-g)));
}
VLO_DEV float gelu_python_bf16(float x) {    // HF GELUActivation(use_gelu_python=True) on bf16
    // x * 0.5 * (1.0 + erf(x / sqrt(2)))  — every op rounds to bf16
    const float a = rbf(x * 0.5f);
    const float t = rbf(x / 1.4142135623730951f);
    const float e = rbf(erff(t));
    const float s = rbf(1.0f + e);
    return rbf(a * s);
}
VLO_DEV float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int KF, int NW, int XSRC, int EPI>
__global__ __launch_bounds__(NW * 64) void gemv16_kernel(GemvArgs a) {
    constexpr int CTG = 2;
    extern __shared__ __attribute__((aligned(16))) float4 red[];      // [2][NW][CTG][64] float4, then scratch
    float *scratch = reinterpret_cast<float *>(red + 2 * NW * CTG * 64);   // rs[16] | tmp[NW*4][16]
    float *rs_lds = scratch;
    float *tmp_lds = scratch + 16;
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int m16 = lane & 15, qd = lane >> 4;
    const int KFtot = a.K >> 5;
    const int kfw0 = (blockIdx.y * NW + w) * a.KC * KF;       // first fragment of this wave's K range

    // ---- group -> tiles -----------------------------------------------------------------
    const int hd = a.kv.head_dim;
    const int tph = (EPI == EPI_ROPE) ? hd / 16 : 2;           // tiles per head
    const int hp = tph / 2;                                    // rotary pairs of tiles per head
    // a.CT == 1: one tile per group (narrow outputs such as o_proj: NT = 256 tiles -> 256 blocks instead of 128)
    const bool single = (EPI != EPI_ROPE && a.CT == 1);
    const int ngroups = (EPI == EPI_ROPE) ? a.NT / 2 : (single ? a.NT : (a.NT + 1) / 2);
    auto tile_a = [&](int g) { return (EPI == EPI_ROPE) ? (g / hp) * tph + (g % hp) : (single ? g : 2 * g); };
    auto tile_b = [&](int g) { return (EPI == EPI_ROPE) ? (g / hp) * tph + (g % hp) + hp : (single ? a.NT : 2 * g + 1); };

    const frag_ab *wbase = reinterpret_cast<const frag_ab *>(a.Wp) + (size_t)kfw0 * 64 + lane;
    const size_t tile_stride = (size_t)KFtot * 64;
    auto item_ptr = [&](int tile, int c) { return wbase + (size_t)tile * tile_stride + (size_t)c * KF * 64; };

    // ---- first weight fragments go in flight before anything else ---------------------------
    frag_ab wr[KF];
    int g = blockIdx.x;
    if (g < ngroups) {
        const frag_ab *wp = item_ptr(tile_a(g), 0);
#pragma unroll
        for (int kf = 0; kf < KF; ++kf) wr[kf] = __builtin_nontemporal_load(wp + kf * 64);
    }

    // ---- XSRC_NORM: rs[m] = rsqrt(mean(h[m]^2) + eps) from the producer's partial sums --------
    if (XSRC == XSRC_NORM) {
        const int t = threadIdx.x, m = t & 15, sl = t >> 4, nsl = NW * 4;
        float s = 0.f;
        for (int p = sl; p < a.sq_in_parts; p += nsl) s += a.sq_in[p * 16 + m];
        tmp_lds[sl * 16 + m] = s;
        __syncthreads();
        if (t < 16) {
            float tot = 0.f;
            for (int i = 0; i < nsl; ++i) tot += tmp_lds[i * 16 + t];
            rs_lds[t] = (t < a.n_rows) ? 1.0f / sqrtf(tot / (float)a.K + a.eps) : 0.f;
        }
        __syncthreads();
    }

    // ---- activation fragments of one K chunk (B operand): x[m = lane&15][k .. k+8] -------------
    frag_ab xf[KF];
    auto load_x = [&](int c) {
        const size_t k0 = (size_t)(kfw0 + c * KF) * 32 + qd * 8;
        const bf16_t *xr = a.x + (size_t)m16 * a.ldx + k0;
        if (XSRC == XSRC_NORM) {
            const float rsm = rs_lds[m16];
#pragma unroll
            for (int kf = 0; kf < KF; ++kf) {
                frag_ab hv = {0, 0, 0, 0, 0, 0, 0, 0}, o;
                if (m16 < a.n_rows) hv = *reinterpret_cast<const frag_ab *>(xr + kf * 32);
                const frag_ab wv = *reinterpret_cast<const frag_ab *>(a.norm_w + k0 + kf * 32);
                unsigned pk[4];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {     // weight * (x * rsqrt(var + eps)).to(bf16)   (HF :66-67)
                    const unsigned t2 = pack2bf(bf2f((bf16_t)hv[j]) * rsm, bf2f((bf16_t)hv[j + 1]) * rsm);
                    pk[j >> 1] = pack2bf(bf2f((bf16_t)wv[j]) * __uint_as_float(t2 << 16),
                                         bf2f((bf16_t)wv[j + 1]) * __uint_as_float(t2 & 0xffff0000u));
                }
                o = __builtin_bit_cast(frag_ab, make_uint4(pk[0], pk[1], pk[2], pk[3]));
                xf[kf] = o;
            }
        } else {
#pragma unroll
            for (int kf = 0; kf < KF; ++kf) {
                frag_ab z = {0, 0, 0, 0, 0, 0, 0, 0};
                if (m16 < a.n_rows) z = *reinterpret_cast<const frag_ab *>(xr + kf * 32);
                xf[kf] = z;
            }
        }
    };
    if (a.KC == 1) load_x(0);

    float sq_acc = 0.f;                    // EPI_RESID: this wave's running sum of squares for row lane&15
    int buf = 0;
    for (; g < ngroups; g += gridDim.x) {
        const int tA = tile_a(g), tB = tile_b(g);
        const bool hasB = tB < a.NT;
        const int gn = g + gridDim.x;
        float4 *rb = red + (size_t)buf * NW * CTG * 64;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < a.KC; ++c) {
            if (a.KC > 1) load_x(c);
            // item (c, A); next item is (c, B) | (c+1, A) | (next group, 0, A) | none
            {
                const frag_ab *np = hasB ? item_ptr(tB, c)
                                         : (c + 1 < a.KC ? item_ptr(tA, c + 1) : (gn < ngroups ? item_ptr(tile_a(gn), 0) : nullptr));
                if (np) {
#pragma unroll
                    for (int kf = 0; kf < KF; ++kf) {
                        acc0 = mfma_bf16(wr[kf], xf[kf], acc0);
                        wr[kf] = __builtin_nontemporal_load(np + kf * 64);
                    }
                } else {
#pragma unroll
                    for (int kf = 0; kf < KF; ++kf) acc0 = mfma_bf16(wr[kf], xf[kf], acc0);
                }
            }
            if (hasB) {
                const frag_ab *np = c + 1 < a.KC ? item_ptr(tA, c + 1) : (gn < ngroups ? item_ptr(tile_a(gn), 0) : nullptr);
                if (np) {
#pragma unroll
                    for (int kf = 0; kf < KF; ++kf) {
                        acc1 = mfma_bf16(wr[kf], xf[kf], acc1);
                        wr[kf] = __builtin_nontemporal_load(np + kf * 64);
                    }
                } else {
#pragma unroll
                    for (int kf = 0; kf < KF; ++kf) acc1 = mfma_bf16(wr[kf], xf[kf], acc1);
                }
            }
        }
        rb[(w * CTG + 0) * 64 + lane] = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
        rb[(w * CTG + 1) * 64 + lane] = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
        __syncthreads();

        // ---- cross-wave reduction + epilogue ------------------------------------------------
        if (EPI == EPI_ROPE) {
            // both tiles of the group are needed by the same lane (rotary pair): one wave
            if (threadIdx.x < 64) {
                const int l = lane;
                float4 sA = make_float4(0, 0, 0, 0), sB = make_float4(0, 0, 0, 0);
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) {
                    sA = f4add(sA, rb[(ww * CTG + 0) * 64 + l]);
                    sB = f4add(sB, rb[(ww * CTG + 1) * 64 + l]);
                }
                const int m = l & 15;
                if (m < a.n_rows) {
                    const float va[4] = {sA.x, sA.y, sA.z, sA.w}, vb[4] = {sB.x, sB.y, sB.z, sB.w};
                    {
                        const int half = hd >> 1, nh = a.num_heads, nkv = a.kv.num_kv_heads;
                        const int head = g / hp, i = (g % hp) * 16 + (l >> 4) * 4;   // column inside the head, < half
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
                                const float x1 = rbf(va[r]), x2 = rbf(vb[r]);      // projection output is bf16
                                const float c = bf2f(cc[r]), s = bf2f(ss[r]);
                                // q*cos + rotate_half(q)*sin, each product and the sum rounded to bf16 (:157-158)
                                lo[r] = f2bf(rbf(x1 * c) + rbf(-x2 * s));
                                hi[r] = f2bf(rbf(x2 * c) + rbf(x1 * s));
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
                }
            }
        } else {
            for (int t = threadIdx.x; t < CTG * 64; t += NW * 64) {
                const int ct = t >> 6, l = t & 63;
                const int tile = ct ? tB : tA;
                float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) s = f4add(s, rb[(ww * CTG + ct) * 64 + l]);
                const int m = l & 15;
                const int col = tile * 16 + (l >> 4) * 4;
                const bool live = (m < a.n_rows) && (tile < a.NT);
                if (EPI == EPI_SWIGLU) {
                    // tile = 8 gate rows (fragment rows 0..7, lanes 0..31) + the 8 up rows of the same columns (lanes 32..63)
                    if (live && l < 32) {
                        float4 u = make_float4(0, 0, 0, 0);
#pragma unroll
                        for (int ww = 0; ww < NW; ++ww) u = f4add(u, rb[(ww * CTG + ct) * 64 + l + 32]);
                        const float gv[4] = {s.x, s.y, s.z, s.w}, uv[4] = {u.x, u.y, u.z, u.w};
                        bf16_t o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = f2bf(silu_bf16(rbf(gv[r])) * rbf(uv[r]));
                        *reinterpret_cast<ushort4 *>(a.out_bf16 + (size_t)m * a.ldo + tile * 8 + (l >> 4) * 4) = *reinterpret_cast<const ushort4 *>(o);
                    }
                } else if (EPI == EPI_RESID) {
                    float sq = 0.f;
                    if (live) {
                        bf16_t *hp4 = a.h + (size_t)m * a.ldo + col;
                        const ushort4 hv = *reinterpret_cast<const ushort4 *>(hp4);
                        const bf16_t hh[4] = {hv.x, hv.y, hv.z, hv.w};
                        const float sv[4] = {s.x, s.y, s.z, s.w};
                        bf16_t o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {      // Linear output -> bf16, then the bf16 residual add
                            const float hn = rbf(bf2f(hh[r]) + rbf(sv[r]));
                            o[r] = f2bf(hn);
                            sq += hn * hn;
                        }
                        *reinterpret_cast<ushort4 *>(hp4) = *reinterpret_cast<const ushort4 *>(o);
                    }
                    sq += __shfl_xor(sq, 16, 64);
                    sq += __shfl_xor(sq, 32, 64);
                    sq_acc += sq;                          // meaningful in lanes 0..15 (row = lane)
                } else if (live) {
                    if (EPI == EPI_PARTIAL_F32) {
                        *reinterpret_cast<float4 *>(a.out_f32 + ((size_t)blockIdx.y * 16 + m) * a.ldo + col) = s;
                    } else if (col < a.N_valid) {          // N padded to 16 at pack time; N_valid % 4 == 0
                        if (a.bias) {
                            const ushort4 b = *reinterpret_cast<const ushort4 *>(a.bias + col);
                            s.x += bf2f(b.x); s.y += bf2f(b.y); s.z += bf2f(b.z); s.w += bf2f(b.w);
                        }
                        ushort4 o;
                        if (EPI == EPI_BF16_GELU_ERF) {
                            o.x = f2bf(gelu_python_bf16(rbf(s.x))); o.y = f2bf(gelu_python_bf16(rbf(s.y)));
                            o.z = f2bf(gelu_python_bf16(rbf(s.z))); o.w = f2bf(gelu_python_bf16(rbf(s.w)));
                        } else {
                            o.x = f2bf(s.x); o.y = f2bf(s.y); o.z = f2bf(s.z); o.w = f2bf(s.w);
                        }
                        *reinterpret_cast<ushort4 *>(a.out_bf16 + (size_t)m * a.ldo + col) = o;
                    }
                }
            }
        }
        buf ^= 1;
    }
    if (EPI == EPI_RESID) {
        // deterministic block total of the row sums of squares -> sq_out[blockIdx.x][16]
        __syncthreads();
        if (lane < 16) tmp_lds[w * 16 + lane] = sq_acc;
        __syncthreads();
        if (threadIdx.x < 16) {
            float tot = 0.f;
            for (int ww = 0; ww < NW; ++ww) tot += tmp_lds[ww * 16 + threadIdx.x];
            a.sq_out[(size_t)blockIdx.x * 16 + threadIdx.x] = tot;
        }
    }
}

// ------------------------------------------------------------------------------------
// host side: plan + launch
// ------------------------------------------------------------------------------------
static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

struct NwKf { int nw, kf; };
// (waves per block, fragments per wave per chunk) combinations that are instantiated, in preference order
static const NwKf kCombos[] = {{8, 16}, {8, 14}, {8, 11}, {8, 8}, {8, 4}, {8, 2}, {8, 1}, {4, 14}, {4, 11}, {4, 1}, {2, 11}, {1, 1}};

int gemv_plan(int K, bool allow_ksplit, GemvPlan *p) {
    if (K <= 0 || (K & 31)) return -1;
    const int KFtot = K >> 5;
    const int force_ks = allow_ksplit ? env_int("VLO_GEMV_KSPLIT", 0) : 0;
    // the combination that keeps the most weight bytes in flight per CU (NW x KF KiB) wins, earlier entries on ties: every
    // model shape gets the same plan as a first-match walk of the list except K = 1792 (Llama-3-8B down-proj at TP = 8),
    // which would otherwise stream with 1 KiB per wave and seven K slices instead of 4 waves x 14 KiB and one
    const NwKf *best = nullptr;
    for (const NwKf &c : kCombos) {
        if (KFtot % (c.nw * c.kf)) continue;
        if (KFtot / (c.nw * c.kf) > 16) continue;
        if (!best || c.nw * c.kf > best->nw * best->kf) best = &c;
    }
    if (!best) return -1;
    const int rest = KFtot / (best->nw * best->kf);   // = KC * ksplit
    int ks = allow_ksplit ? rest : 1;                // K slices across blocks (fp32 partial outputs) vs chunks inside a wave
    if (allow_ksplit && force_ks > 0 && rest % force_ks == 0) ks = force_ks;
    p->NW = best->nw; p->KF = best->kf; p->ksplit = ks; p->KC = rest / ks;
    return 0;
}

// groups of two column tiles, or single tiles when that balances better over the CUs (pairs are mandatory for the
// rotary epilogue).  e.g. gate/up: 1792 tiles = 7.0 per CU as singles, 896 pairs = 3.5 -> 4 per CU (87.5 %).
static bool single_tile_groups(const GemvArgs &a, const GemvPlan &p, int epi) {
    if (epi == EPI_ROPE) return false;
    const int slots = 256 / p.ksplit > 0 ? 256 / p.ksplit : 1;
    auto eff = [&](int items) {
        const int rounds = (items + slots - 1) / slots;
        return (double)items / ((double)rounds * slots);
    };
    return eff(a.NT) > eff((a.NT + 1) / 2) + 0.2;     // HBM saturates below 256 CUs: a 12 % idle-CU tail costs less than 2x the barriers
}
static int groups_of(const GemvArgs &a, const GemvPlan &p, int epi) {
    if (epi == EPI_ROPE) return a.NT / 2;
    return single_tile_groups(a, p, epi) ? a.NT : (a.NT + 1) / 2;
}

int gemv_grid_x(const GemvArgs &a, const GemvPlan &p, int epi) {
    static const int kBPC = env_int("VLO_GEMV_BPC", 1);          // resident blocks per CU aimed at
    static const int kCUs = 256;
    const int ngroups = groups_of(a, p, epi);
    int gx = (kCUs * kBPC) / p.ksplit;
    if (gx < 1) gx = 1;
    if (gx > ngroups) gx = ngroups;
    const int per = (ngroups + gx - 1) / gx;        // balance: same number of groups per block where possible
    return (ngroups + per - 1) / per;
}

template <int KF, int NW>
static hipError_t launch_variant(const GemvArgs &a, int xsrc, int epi, dim3 grid, size_t lds, hipStream_t st) {
    dim3 block(NW * 64);
#define VLO_GO(XS, EP)                                                                    \
    do {                                                                                  \
        hipLaunchKernelGGL((gemv16_kernel<KF, NW, XS, EP>), grid, block, lds, st, a);     \
        return hipGetLastError();                                                         \
    } while (0)
    if (xsrc == XSRC_NORM) {
        if (epi == EPI_SWIGLU) VLO_GO(XSRC_NORM, EPI_SWIGLU);
        if (epi == EPI_ROPE) VLO_GO(XSRC_NORM, EPI_ROPE);      // run_chunk_fused: input RMSNorm on the qkv operand load
        if (epi == EPI_BF16) VLO_GO(XSRC_NORM, EPI_BF16);      // run_chunk_fused: final RMSNorm on the lm_head operand load
    } else {
        if (epi == EPI_ROPE) VLO_GO(XSRC_PLAIN, EPI_ROPE);
        if (epi == EPI_SWIGLU) VLO_GO(XSRC_PLAIN, EPI_SWIGLU);
        if (epi == EPI_RESID) VLO_GO(XSRC_PLAIN, EPI_RESID);
        if (epi == EPI_BF16) VLO_GO(XSRC_PLAIN, EPI_BF16);
        if (epi == EPI_BF16_GELU_ERF) VLO_GO(XSRC_PLAIN, EPI_BF16_GELU_ERF);
        if (epi == EPI_PARTIAL_F32) VLO_GO(XSRC_PLAIN, EPI_PARTIAL_F32);
    }
#undef VLO_GO
    return hipErrorInvalidValue;
}

hipError_t gemv_launch(GemvArgs a, const GemvPlan &p, int xsrc, int epi, hipStream_t st) {
    if (epi != EPI_PARTIAL_F32 && p.ksplit != 1) return hipErrorInvalidValue;
    if (epi == EPI_ROPE && ((a.NT & 1) || (a.kv.head_dim != 64 && a.kv.head_dim != 128))) return hipErrorInvalidValue;
    a.CT = single_tile_groups(a, p, epi) ? 1 : 2;
    a.KC = p.KC;
    dim3 grid(gemv_grid_x(a, p, epi), p.ksplit);
    const size_t lds = (size_t)2 * p.NW * 2 * 64 * sizeof(float4) + (16 + (size_t)p.NW * 4 * 16) * sizeof(float);
#define VLO_CASE(NW_, KF_) \
    if (p.NW == NW_ && p.KF == KF_) return launch_variant<KF_, NW_>(a, xsrc, epi, grid, lds, st);
    VLO_CASE(8, 16) VLO_CASE(8, 14) VLO_CASE(8, 11) VLO_CASE(8, 8) VLO_CASE(8, 4) VLO_CASE(8, 2) VLO_CASE(8, 1)
    VLO_CASE(4, 14) VLO_CASE(4, 11) VLO_CASE(4, 1) VLO_CASE(2, 11) VLO_CASE(1, 1)
#undef VLO_CASE
    return hipErrorInvalidValue;
}

hipError_t pack_weight_launch(const void *W, void *Wp, int N_valid, int K, int ldw, int NT, int tile_stride, int tile_offset,
                              int half, hipStream_t st) {
    const int KFtot = K >> 5;
    const size_t total = (size_t)NT * KFtot * 64;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, st, (const bf16_t *)W, (uint4 *)Wp, N_valid, K, ldw, NT,
                       KFtot, tile_stride, tile_offset, half);
    return hipGetLastError();
}
