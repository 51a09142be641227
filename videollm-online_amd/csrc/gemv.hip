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

// One (tile, lane) element of the reduced output -> epilogue.  s = sum over the block's waves.
template <int EPI>
VLO_DEV void gemv_epilogue(const GemvArgs &a, int tile, int l, float4 s, float4 s2, int kslice) {
    const int m = l & 15;
    if (m >= a.n_rows) return;
    if (EPI == EPI_SWIGLU) {           // s = gate tile, s2 = up tile; `tile` is the pair index
        const int col = tile * 16 + (l >> 4) * 4;
        ushort4 o;
        o.x = f2bf(silu_bf16(rbf(s.x)) * rbf(s2.x));
        o.y = f2bf(silu_bf16(rbf(s.y)) * rbf(s2.y));
        o.z = f2bf(silu_bf16(rbf(s.z)) * rbf(s2.z));
        o.w = f2bf(silu_bf16(rbf(s.w)) * rbf(s2.w));
        *reinterpret_cast<ushort4 *>(a.out_bf16 + (size_t)m * a.ldo + col) = o;
        return;
    }
    const int col = tile * 16 + (l >> 4) * 4;
    if (EPI == EPI_PARTIAL_F32) {
        *reinterpret_cast<float4 *>(a.out_f32 + ((size_t)kslice * 16 + m) * a.ldo + col) = s;
        return;
    }
    if (col >= a.N_valid) return;          // N padded to 16 at pack time; N_valid % 4 == 0
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

// Persistent, software-pipelined weight streamer.
//   grid.y = K slices; grid.x blocks stride over groups of CTG consecutive column tiles.
//   Wave w of a block owns KF weight fragments (K range) of every tile the block visits and keeps
//   the matching activation fragments in VGPRs.  The loads of tile t+1 are issued before the MFMAs
//   of tile t (two register sets), also across group boundaries, so the wave always has 16..32 KiB
//   of HBM reads in flight.  Per group: partial tiles -> LDS (double-buffered), one barrier, the
//   first CTG*64 threads reduce across waves and run the epilogue.
template <int KF, int NW, int EPI>
__global__ __launch_bounds__(NW * 64) void gemv16_kernel(GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 red[];      // [2][NW][CTG][64]
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int KFtot = a.K >> 5;
    const int kf0 = (blockIdx.y * NW + w) * KF;
    const int CTG = a.CT;

    frag_ab xf[KF];
    {
        const bf16_t *xr = a.x + (size_t)(lane & 15) * a.ldx + (size_t)kf0 * 32 + (lane >> 4) * 8;
#pragma unroll
        for (int kf = 0; kf < KF; ++kf) xf[kf] = *reinterpret_cast<const frag_ab *>(xr + kf * 32);
    }
    const frag_ab *wbase = reinterpret_cast<const frag_ab *>(a.Wp) + (size_t)kf0 * 64 + lane;
    const size_t tile_stride = (size_t)KFtot * 64;
    const int ngroups = (a.NT + CTG - 1) / CTG;

    // rolling prefetch: w[kf] is re-loaded with the next tile's fragment right after the MFMA that
    // consumed it, so KF x 1 KiB of HBM reads stay in flight per wave with one register set.
    frag_ab wr[KF];
    int g = blockIdx.x;
    if (g < ngroups) {
        const frag_ab *wp = wbase + (size_t)(g * CTG) * tile_stride;
#pragma unroll
        for (int kf = 0; kf < KF; ++kf) wr[kf] = __builtin_nontemporal_load(wp + kf * 64);
    }
    int buf = 0;
    for (; g < ngroups; g += gridDim.x) {
        const int tile0 = g * CTG;
        const int cnt = min(CTG, a.NT - tile0);
        float4 *rb = red + (size_t)buf * NW * CTG * 64;
        for (int ct = 0; ct < cnt; ++ct) {
            // the next tile this wave will need (next in group, or first of the block's next group)
            int nt = tile0 + ct + 1;
            if (ct + 1 == cnt) nt = (g + gridDim.x < ngroups) ? (g + gridDim.x) * CTG : -1;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (nt >= 0) {
                const frag_ab *wp = wbase + (size_t)nt * tile_stride;
#pragma unroll
                for (int kf = 0; kf < KF; ++kf) {
                    acc = mfma_bf16(wr[kf], xf[kf], acc);
                    wr[kf] = __builtin_nontemporal_load(wp + kf * 64);
                }
            } else {
#pragma unroll
                for (int kf = 0; kf < KF; ++kf) acc = mfma_bf16(wr[kf], xf[kf], acc);
            }
            rb[(w * CTG + ct) * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        __syncthreads();
        if (EPI == EPI_SWIGLU) {
            const int npair = cnt >> 1;
            for (int t = threadIdx.x; t < npair * 64; t += NW * 64) {
                const int pr = t >> 6, l = t & 63;
                float4 gsum = make_float4(0, 0, 0, 0), usum = make_float4(0, 0, 0, 0);
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) {
                    const float4 a0 = rb[(ww * CTG + 2 * pr) * 64 + l];
                    const float4 a1 = rb[(ww * CTG + 2 * pr + 1) * 64 + l];
                    gsum.x += a0.x; gsum.y += a0.y; gsum.z += a0.z; gsum.w += a0.w;
                    usum.x += a1.x; usum.y += a1.y; usum.z += a1.z; usum.w += a1.w;
                }
                gemv_epilogue<EPI>(a, (tile0 >> 1) + pr, l, gsum, usum, blockIdx.y);
            }
        } else {
            for (int t = threadIdx.x; t < cnt * 64; t += NW * 64) {
                const int ct = t >> 6, l = t & 63;
                float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) {
                    const float4 v = rb[(ww * CTG + ct) * 64 + l];
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
                gemv_epilogue<EPI>(a, tile0 + ct, l, s, s, blockIdx.y);
            }
        }
        buf ^= 1;
    }
}

// ------------------------------------------------------------------------------------
// host side: plan + launch
// ------------------------------------------------------------------------------------
#include <stdlib.h>

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

struct NwKf { int nw, kf; };
// (waves per block, fragments per wave) combinations that are instantiated, in preference order
static const NwKf kCombos[] = {{8, 16}, {8, 14}, {8, 11}, {8, 8}, {4, 16}, {4, 14}, {4, 11}, {8, 4}, {4, 8},
                               {8, 2}, {4, 4}, {2, 11}, {8, 1}, {4, 2}, {2, 8}, {2, 4}, {4, 1}, {2, 2}, {2, 1}, {1, 1}};

int gemv_plan(int K, bool allow_ksplit, GemvPlan *p) {
    if (K <= 0 || (K & 31)) return -1;
    const int KFtot = K >> 5;
    const int force_nw = env_int("VLO_GEMV_NW", 0), force_kf = env_int("VLO_GEMV_KF", 0);
    for (const NwKf &c : kCombos) {
        if (force_nw && c.nw != force_nw) continue;
        if (force_kf && c.kf != force_kf) continue;
        if (KFtot % (c.nw * c.kf)) continue;
        const int ks = KFtot / (c.nw * c.kf);
        if (ks > 1 && !allow_ksplit) continue;
        if (ks > 16) continue;
        p->NW = c.nw; p->KF = c.kf; p->ksplit = ks;
        return 0;
    }
    return -1;
}

template <int KF, int NW>
static hipError_t launch_epi(const GemvArgs &a, int epi, dim3 grid, size_t lds, hipStream_t st) {
    dim3 block(NW * 64);
    switch (epi) {
    case EPI_PARTIAL_F32: hipLaunchKernelGGL((gemv16_kernel<KF, NW, EPI_PARTIAL_F32>), grid, block, lds, st, a); break;
    case EPI_BF16: hipLaunchKernelGGL((gemv16_kernel<KF, NW, EPI_BF16>), grid, block, lds, st, a); break;
    case EPI_BF16_GELU_ERF: hipLaunchKernelGGL((gemv16_kernel<KF, NW, EPI_BF16_GELU_ERF>), grid, block, lds, st, a); break;
    case EPI_SWIGLU: hipLaunchKernelGGL((gemv16_kernel<KF, NW, EPI_SWIGLU>), grid, block, lds, st, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t gemv_launch(GemvArgs a, const GemvPlan &p, int epi, hipStream_t st) {
    static const int kCTG = env_int("VLO_GEMV_CTG", 2);
    static const int kBPC = env_int("VLO_GEMV_BPC", 1);          // resident blocks per CU aimed at
    static const int kCUs = 256;
    int ctg = a.CT > 0 ? a.CT : kCTG;
    if (epi == EPI_SWIGLU) ctg = (ctg + 1) & ~1;
    a.CT = ctg;
    const int ngroups = (a.NT + ctg - 1) / ctg;
    int gx = (kCUs * kBPC) / p.ksplit;
    if (gx < 1) gx = 1;
    if (gx > ngroups) gx = ngroups;
    // balance: every block should get the same number of groups where possible
    const int per = (ngroups + gx - 1) / gx;
    gx = (ngroups + per - 1) / per;
    dim3 grid(gx, p.ksplit);
    const size_t lds = (size_t)2 * p.NW * ctg * 64 * sizeof(float4);
#define VLO_CASE(NW_, KF_) \
    if (p.NW == NW_ && p.KF == KF_) return launch_epi<KF_, NW_>(a, epi, grid, lds, st);
    VLO_CASE(8, 16) VLO_CASE(8, 14) VLO_CASE(8, 11) VLO_CASE(8, 8) VLO_CASE(4, 16) VLO_CASE(4, 14) VLO_CASE(4, 11)
    VLO_CASE(8, 4) VLO_CASE(4, 8) VLO_CASE(8, 2) VLO_CASE(4, 4) VLO_CASE(2, 11) VLO_CASE(8, 1) VLO_CASE(4, 2)
    VLO_CASE(2, 8) VLO_CASE(2, 4) VLO_CASE(4, 1) VLO_CASE(2, 2) VLO_CASE(2, 1) VLO_CASE(1, 1)
#undef VLO_CASE
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
