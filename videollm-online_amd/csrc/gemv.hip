// gemv.hip — weight-streaming skinny GEMM for the Llama step (n <= 16 rows).
//
// y[m][n] = sum_k x[m][k] * W[n][k]      x: bf16 [16][ldx] row-major (rows >= n_rows ignored)
//                                        W: bf16, HF nn.Linear layout [N][K]
//
// Replaces the hipBLASLt/cuBLAS skinny GEMMs under q/k/v/o_proj, gate/up/down_proj,
// lm_head (HF:models/llama/modeling_llama.py:174-176,254-256,280,477-480) and the
// connector Linears (models/live_llama/modeling_live_llama.py:18-22).
//
// HBM-bound (arithmetic intensity ~ n FLOP per weight byte): the design streams every
// weight byte exactly once with 1-KiB-per-wave coalesced global_load_dwordx4 and keeps
// the tiny activation operand in registers.
//
// Weight image in HBM ("packed", built once at load time by pack_weight_kernel):
//   Wp[tile][kf][lane] : 16 bytes = W[tile*16 + (lane&15)][kf*32 + (lane>>4)*8 .. +8]
// i.e. exactly the A-operand fragment of v_mfma_f32_16x16x32_bf16, so one wave
// instruction loads one MFMA's worth of weights from 1 KiB of contiguous HBM.
// The MFMA computes D[nrow][m] = sum_k Wfrag[nrow][k] * xfrag[k][m]   (W as A, x as B):
//   lane l, reg r  ->  output column n = tile*16 + (l>>4)*4 + r, token row m = l&15.
//
// Work split: grid.x = groups of CT column tiles, grid.y = K splits; inside a block the
// NW waves split the block's K range (wave w owns KF fragments), each wave keeps its x
// fragments in VGPRs for all CT tiles, partial tiles are reduced across waves through LDS.
#include "common.cuh"
#include "gemv.h"

// ------------------------------------------------------------------------------------
// packing: row-major [N][K] bf16 -> fragment order.  Rows >= N_valid are zero-filled.
// ------------------------------------------------------------------------------------
// Source tile t lands at destination tile t*tile_stride + tile_offset (used to concatenate q/k/v and
// to interleave gate/up 16-row tiles for the SwiGLU epilogue).
__global__ void pack_weight_kernel(const bf16_t *__restrict__ W, uint4 *__restrict__ Wp, int N_valid, int K,
                                   int NT, int KFtot, int tile_stride, int tile_offset) {
    const size_t total = (size_t)NT * KFtot * 64;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t t = i >> 6;
        const int kf = (int)(t % KFtot);
        const int tile = (int)(t / KFtot);
        const int n = tile * 16 + (lane & 15);
        const int k = kf * 32 + (lane >> 4) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (n < N_valid) v = *reinterpret_cast<const uint4 *>(W + (size_t)n * K + k);
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

template <int KF, int EPI>
__global__ __launch_bounds__(1024) void gemv16_kernel(GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 red[];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int NW = blockDim.x >> 6;
    const int KFtot = a.K >> 5;
    const int kf0 = (blockIdx.y * NW + w) * KF;

    // activation fragments (B operand): x[m = lane&15][k = (kf0+kf)*32 + (lane>>4)*8 ..]
    frag_ab xf[KF];
    {
        const bf16_t *xr = a.x + (size_t)(lane & 15) * a.ldx + (size_t)kf0 * 32 + (lane >> 4) * 8;
#pragma unroll
        for (int kf = 0; kf < KF; ++kf) xf[kf] = *reinterpret_cast<const frag_ab *>(xr + kf * 32);
    }
    const int tile0 = blockIdx.x * a.CT;
    const int ntiles = min(a.CT, a.NT - tile0);
    for (int ct = 0; ct < ntiles; ++ct) {
        const frag_ab *wp = reinterpret_cast<const frag_ab *>(a.Wp) + ((size_t)(tile0 + ct) * KFtot + kf0) * 64 + lane;
        frag_ab wf[KF];
#pragma unroll
        for (int kf = 0; kf < KF; ++kf) wf[kf] = __builtin_nontemporal_load(wp + kf * 64);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kf = 0; kf < KF; ++kf) acc = mfma_bf16(wf[kf], xf[kf], acc);
        red[(w * a.CT + ct) * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    __syncthreads();

    // cross-wave reduction + epilogue: thread t owns (tile ct = t/64, fragment lane l = t%64)
    if (EPI == EPI_SWIGLU) {
        // tiles come in (gate, up) pairs
        const int npair = ntiles >> 1;
        for (int t = threadIdx.x; t < npair * 64; t += blockDim.x) {
            const int pr = t >> 6, l = t & 63;
            float4 g = make_float4(0, 0, 0, 0), u = make_float4(0, 0, 0, 0);
            for (int ww = 0; ww < NW; ++ww) {
                const float4 a0 = red[(ww * a.CT + 2 * pr) * 64 + l];
                const float4 a1 = red[(ww * a.CT + 2 * pr + 1) * 64 + l];
                g.x += a0.x; g.y += a0.y; g.z += a0.z; g.w += a0.w;
                u.x += a1.x; u.y += a1.y; u.z += a1.z; u.w += a1.w;
            }
            const int m = l & 15;
            if (m < a.n_rows) {
                const int col = ((tile0 >> 1) + pr) * 16 + (l >> 4) * 4;    // column in the [.., I] activation
                ushort4 o;
                o.x = f2bf(silu_bf16(rbf(g.x)) * rbf(u.x));
                o.y = f2bf(silu_bf16(rbf(g.y)) * rbf(u.y));
                o.z = f2bf(silu_bf16(rbf(g.z)) * rbf(u.z));
                o.w = f2bf(silu_bf16(rbf(g.w)) * rbf(u.w));
                *reinterpret_cast<ushort4 *>(a.out_bf16 + (size_t)m * a.ldo + col) = o;
            }
        }
        return;
    }
    for (int t = threadIdx.x; t < ntiles * 64; t += blockDim.x) {
        const int ct = t >> 6, l = t & 63;
        float4 s = make_float4(0, 0, 0, 0);
        for (int ww = 0; ww < NW; ++ww) {
            const float4 v = red[(ww * a.CT + ct) * 64 + l];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int m = l & 15;
        if (m >= a.n_rows) continue;
        const int col = (tile0 + ct) * 16 + (l >> 4) * 4;
        if (EPI == EPI_PARTIAL_F32) {
            *reinterpret_cast<float4 *>(a.out_f32 + ((size_t)blockIdx.y * 16 + m) * a.ldo + col) = s;
        } else {
            if (col >= a.N_valid) continue;       // N padded to 16 at pack time; N_valid % 4 == 0
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

// ------------------------------------------------------------------------------------
// host side: plan + launch
// ------------------------------------------------------------------------------------
static const int kKFSet[] = {16, 14, 11, 8, 4, 2, 1};

int gemv_plan(int K, bool allow_ksplit, GemvPlan *p) {
    if (K <= 0 || (K & 31)) return -1;
    const int KFtot = K >> 5;
    // prefer many waves per block (deep load queues), then big KF, then a K split for long K
    int best_score = -1;
    for (int nw = 16; nw >= 1; nw >>= 1) {
        for (int kf : kKFSet) {
            if (KFtot % (nw * kf)) continue;
            const int ks = KFtot / (nw * kf);
            if (ks > 1 && !allow_ksplit) continue;
            if (ks > 16) continue;
            // score: 8 waves x KF in [8,16] is the sweet spot measured for K=4096/14336
            int score = 0;
            score += (nw == 8) ? 40 : (nw == 4 ? 25 : (nw == 16 ? 20 : (nw == 2 ? 10 : 0)));
            score += (kf >= 8) ? 30 + kf : kf;
            score -= 3 * (ks - 1);
            if (score > best_score) {
                best_score = score;
                p->NW = nw; p->KF = kf; p->ksplit = ks;
            }
        }
    }
    return best_score < 0 ? -1 : 0;
}

template <int EPI>
static hipError_t launch_kf(const GemvArgs &a, const GemvPlan &p, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    switch (p.KF) {
#define VLO_CASE(KF_) \
    case KF_: hipLaunchKernelGGL((gemv16_kernel<KF_, EPI>), grid, block, lds, st, a); break;
        VLO_CASE(16) VLO_CASE(14) VLO_CASE(11) VLO_CASE(8) VLO_CASE(4) VLO_CASE(2) VLO_CASE(1)
#undef VLO_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t gemv_launch(GemvArgs a, const GemvPlan &p, int epi, hipStream_t st) {
    // CT: column tiles per block — aim at ~4 blocks per CU worth of blocks, cap LDS at 64 KiB
    int ct = a.CT;
    if (ct <= 0) {
        const int target_blocks = 1024;
        ct = (a.NT * p.ksplit + target_blocks - 1) / target_blocks;
        if (ct < 1) ct = 1;
        const int ct_max = 32 / p.NW > 0 ? 32 / p.NW : 1;      // NW*CT KiB of LDS (<= 32 KiB)
        if (ct > ct_max) ct = ct_max;
        if (epi == EPI_SWIGLU) ct = (ct + 1) & ~1;
    }
    a.CT = ct;
    dim3 grid((a.NT + ct - 1) / ct, p.ksplit), block(p.NW * 64);
    const size_t lds = (size_t)p.NW * ct * 64 * sizeof(float4);
    switch (epi) {
    case EPI_PARTIAL_F32: return launch_kf<EPI_PARTIAL_F32>(a, p, grid, block, lds, st);
    case EPI_BF16: return launch_kf<EPI_BF16>(a, p, grid, block, lds, st);
    case EPI_BF16_GELU_ERF: return launch_kf<EPI_BF16_GELU_ERF>(a, p, grid, block, lds, st);
    case EPI_SWIGLU: return launch_kf<EPI_SWIGLU>(a, p, grid, block, lds, st);
    }
    return hipErrorInvalidValue;
}

hipError_t pack_weight_launch(const void *W, void *Wp, int N_valid, int K, int NT, int tile_stride, int tile_offset,
                              hipStream_t st) {
    const int KFtot = K >> 5;
    const size_t total = (size_t)NT * KFtot * 64;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, st, (const bf16_t *)W, (uint4 *)Wp, N_valid, K, NT,
                       KFtot, tile_stride, tile_offset);
    return hipGetLastError();
}
