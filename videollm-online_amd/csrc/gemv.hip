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
// so a decoder layer is 7 launches (add_rmsnorm, qkv, attention, combine, o, gate_up, down; engine.hip::run_chunk) instead of ~14.
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

// fp8 e4m3 image (vlo_config.weight_dtype = 1): half the bytes per weight, the same one-16-byte-load-per-lane stream:
//   Wp8[tile][kf2][lane] : 16 bytes = W[tile*16 + (lane&15)][(2*kf2)*32 + (lane>>4)*8 .. +8]  ++  the same 8 of fragment 2*kf2+1
// so one load feeds TWO MFMAs (expanded to bf16 in registers, exactly); the per-output-channel scale is applied to the
// reduced fp32 sums in the epilogue, in packed row order.
__global__ void pack_weight_fp8_kernel(const uint8_t *__restrict__ W, uint4 *__restrict__ Wp, int N_valid, int K, int ldw,
                                       int NT, int KF2tot, int tile_stride, int tile_offset, int half) {
    const size_t total = (size_t)NT * KF2tot * 64;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t t = i >> 6;
        const int kf2 = (int)(t % KF2tot);
        const int tile = (int)(t / KF2tot);
        const int k = kf2 * 64 + (lane >> 4) * 8;
        const int r16 = lane & 15;
        int n;
        if (half < 0) n = tile * 16 + r16;
        else if ((r16 >> 3) != half) continue;
        else n = tile * 8 + (r16 & 7);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (n < N_valid) {
            const uint2 lo = *reinterpret_cast<const uint2 *>(W + (size_t)n * ldw + k);
            const uint2 hi = *reinterpret_cast<const uint2 *>(W + (size_t)n * ldw + k + 32);
            v = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        Wp[((size_t)(tile * tile_stride + tile_offset) * KF2tot + kf2) * 64 + lane] = v;
    }
}
__global__ void pack_scale_kernel(const float *__restrict__ scale, float *__restrict__ dst, int N_valid, int NT, int tile_stride,
                                  int tile_offset, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NT * 16) return;
    const int tile = i >> 4, r16 = i & 15;
    int n;
    if (half < 0) n = tile * 16 + r16;
    else if ((r16 >> 3) != half) return;
    else n = tile * 8 + (r16 & 7);
    dst[(size_t)(tile * tile_stride + tile_offset) * 16 + r16] = n < N_valid ? scale[n] : 0.f;
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
VLO_DEV float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

template <int KF, int NW, int XSRC, int EPI, int WQ, int GB = 1>
__global__ __launch_bounds__(NW * 64) void gemv16_kernel(GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 red[];      // [2][NW][CTG][64] float4, then scratch
#define VLO_GEMV_BX blockIdx.x
#define VLO_GEMV_BY blockIdx.y
#define VLO_GEMV_GX gridDim.x
#include "gemv_body.inc"
#undef VLO_GEMV_BX
#undef VLO_GEMV_BY
#undef VLO_GEMV_GX
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
    const int force_ks = 0;
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
    constexpr int kBPC = 1;                                      // resident blocks per CU (2 / 3 measured slower: profiles/r3_gemv_blocks_per_cu_sweep.txt)
    static const int kCUs = env_int("VLO_GEMV_CUS", 256);        // (tests shrink the grid so that a block walks many groups)
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
    // K chunks walked sequentially (KC > 1: K = 8192 at 8 waves x 16 fragments) take the chunk-outer batch loop (gemv_body.inc): batches of
    // 4 groups, 2 for the bf16 norm-on-load kernel (4 would spill: 16 + 16 fragment registers + the normalisation's temporaries).  Measured on the
    // 70B shapes (MI355X, profiles/r4_gemv_batch_loop.txt): fp8 gate/up 121.5 -> 87.9 us, qkv 26.2 -> 21.1, lm_head 184 -> 171; bf16 gate/up
    // 161.9 -> 155.4, qkv 36.6 -> 33.6; the bf16 lm_head (15.7 groups per block, the longest steady state) 313.6 -> 324.5: left on the group-outer loop
    static const int kBatch = env_int("VLO_GEMV_BATCH", 1);
#define VLO_GO(XS, EP)                                                                            \
    do {                                                                                          \
        constexpr bool kHasBatch = (KF == 16 && NW == 8);                                         \
        const bool batch = kHasBatch && kBatch && a.KC > 1 && (EP) != EPI_BF16_GELU_ERF; \
        if (a.wq) {                                                                               \
            if constexpr ((KF & 1) == 0 && (NW == 8 || (NW == 4 && KF == 14))) {    /* 4 x 14: K = 1792, the 8B down-proj shard at TP = 8 */ \
                if constexpr (kHasBatch && (EP) != EPI_BF16_GELU_ERF) { \
                    if (batch) { hipLaunchKernelGGL((gemv16_kernel<KF, NW, XS, EP, 1, 4>), grid, block, lds, st, a); return hipGetLastError(); } \
                }                                                                                 \
                hipLaunchKernelGGL((gemv16_kernel<KF, NW, XS, EP, 1>), grid, block, lds, st, a);  \
            } else {                                                                              \
                return hipErrorInvalidValue;      /* fp8 image: two fragments per 16-byte load */ \
            }                                                                                     \
        } else {                                                                                  \
            if constexpr (kHasBatch && (EP) != EPI_BF16_GELU_ERF && (EP) != EPI_BF16) {    \
                if (batch) { hipLaunchKernelGGL((gemv16_kernel<KF, NW, XS, EP, 0, ((XS) == XSRC_NORM ? 2 : 4)>), grid, block, lds, st, a); return hipGetLastError(); } \
            }                                                                                     \
            hipLaunchKernelGGL((gemv16_kernel<KF, NW, XS, EP, 0>), grid, block, lds, st, a);      \
        }                                                                                         \
        return hipGetLastError();                                                                 \
    } while (0)
    if (xsrc == XSRC_NORM) {
        if (epi == EPI_SWIGLU) VLO_GO(XSRC_NORM, EPI_SWIGLU);
    } else {
        if (epi == EPI_ROPE) VLO_GO(XSRC_PLAIN, EPI_ROPE);
        if (epi == EPI_SWIGLU) VLO_GO(XSRC_PLAIN, EPI_SWIGLU);
        if (epi == EPI_RESID) VLO_GO(XSRC_PLAIN, EPI_RESID);
        if (epi == EPI_BF16) VLO_GO(XSRC_PLAIN, EPI_BF16);
        if (epi == EPI_BF16_GELU_ERF) VLO_GO(XSRC_PLAIN, EPI_BF16_GELU_ERF);
        if (epi == EPI_PARTIAL_F32) VLO_GO(XSRC_PLAIN, EPI_PARTIAL_F32);
        if (epi == EPI_PARTIAL_MBOX) VLO_GO(XSRC_PLAIN, EPI_PARTIAL_MBOX);
    }
#undef VLO_GO
    return hipErrorInvalidValue;
}

hipError_t gemv_prepare(GemvArgs *a, const GemvPlan &p, int epi, int *grid_x, int *grid_y, size_t *lds_bytes) {
    if (epi != EPI_PARTIAL_F32 && p.ksplit != 1) return hipErrorInvalidValue;
    if (epi == EPI_PARTIAL_MBOX && (a->mbox_T < 1 || a->mbox_T > 8 || !a->mbox[0])) return hipErrorInvalidValue;
    if (epi == EPI_ROPE && ((a->NT & 1) || (a->kv.head_dim != 64 && a->kv.head_dim != 128))) return hipErrorInvalidValue;
    a->CT = single_tile_groups(*a, p, epi) ? 1 : 2;
    a->KC = p.KC;
    *grid_x = gemv_grid_x(*a, p, epi);
    *grid_y = p.ksplit;
    *lds_bytes = (size_t)2 * p.NW * 2 * 64 * sizeof(float4) + (16 + (size_t)p.NW * 4 * 16) * sizeof(float);
    return hipSuccess;
}

hipError_t gemv_launch(GemvArgs a, const GemvPlan &p, int xsrc, int epi, hipStream_t st) {
    int gx = 0, gy = 0;
    size_t lds = 0;
    const hipError_t pe = gemv_prepare(&a, p, epi, &gx, &gy, &lds);
    if (pe != hipSuccess) return pe;
    // the kernel's first weight-fragment and activation-row loads are UNCONDITIONAL (gemv_head.inc, gemv_body.inc): every block must own at
    // least one group and there must be a first row to read — an oversized grid or an empty input would read past the image instead of idling
    if (a.n_rows < 1 || a.n_rows > 16 || gx < 1 || gx > groups_of(a, p, epi)) return hipErrorInvalidValue;
    dim3 grid(gx, gy);
#define VLO_CASE(NW_, KF_) \
    if (p.NW == NW_ && p.KF == KF_) return launch_variant<KF_, NW_>(a, xsrc, epi, grid, lds, st);
    VLO_CASE(8, 16) VLO_CASE(8, 14) VLO_CASE(8, 11) VLO_CASE(8, 8) VLO_CASE(8, 4) VLO_CASE(8, 2) VLO_CASE(8, 1)
    VLO_CASE(4, 14) VLO_CASE(4, 11) VLO_CASE(4, 1) VLO_CASE(2, 11) VLO_CASE(1, 1)
#undef VLO_CASE
    return hipErrorInvalidValue;
}

hipError_t pack_weight_fp8_launch(const void *W, const float *scale, void *Wp, float *scale_p, int N_valid, int K, int ldw, int NT,
                                  int tile_stride, int tile_offset, int half, hipStream_t st) {
    if (K & 63) return hipErrorInvalidValue;
    const int KF2tot = K >> 6;
    const size_t total = (size_t)NT * KF2tot * 64;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(pack_weight_fp8_kernel, dim3(blocks), dim3(256), 0, st, (const uint8_t *)W, (uint4 *)Wp, N_valid, K, ldw, NT, KF2tot,
                       tile_stride, tile_offset, half);
    hipLaunchKernelGGL(pack_scale_kernel, dim3((NT * 16 + 255) / 256), dim3(256), 0, st, scale, scale_p, N_valid, NT, tile_stride,
                       tile_offset, half);
    return hipGetLastError();
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
