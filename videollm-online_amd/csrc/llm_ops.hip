// llm_ops.hip — the non-GEMV kernels of one Llama streaming step on gfx950.
//
//   prep_rows_kernel        stage the step's input embeddings + their row sums of squares
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
#include "common.cuh"
#include "llm_ops.h"

// ------------------------------------------------------------------------------------
// step input: copy the n new embedding rows into the residual stream and emit each row's sum of
// squares (consumed by the first layer's norm-on-load GEMV).  One block per row.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_rows_kernel(const bf16_t *__restrict__ src, bf16_t *__restrict__ h,
                                                        float *__restrict__ sq_out, int H) {
    __shared__ float sm[16];
    const int m = blockIdx.x;
    float ss = 0.f;
    for (int ch = threadIdx.x; ch < (H >> 3); ch += blockDim.x) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(src + (size_t)m * H + ch * 8);
        *reinterpret_cast<uint4 *>(h + (size_t)m * H + ch * 8) = raw;
        const bf16_t *e = reinterpret_cast<const bf16_t *>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += bf2f(e[j]) * bf2f(e[j]);
    }
    ss = block_sum(ss, sm);
    if (threadIdx.x == 0) sq_out[m] = ss;          // sq_out[0][m]: a single partial
}
hipError_t prep_rows_launch(const unsigned short *src, unsigned short *h, float *sq_out, int rows, int H, hipStream_t st) {
    if (H & 7) return hipErrorInvalidValue;
    hipLaunchKernelGGL(prep_rows_kernel, dim3(rows), dim3(256), 0, st, src, h, sq_out, H);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// chunk attention.  grid = (nsplit, nkv); block = (G / HPW) waves; wave w owns q heads
// kvh*G + w*HPW .. +HPW and walks the block's key chunk in 32-key steps:
//   S^T[key][qrow] = K[key][:] . Q[qrow][:]          (K tile as MFMA A operand, from HBM)
//   online softmax per (head, qrow = lane&15); P^T stays in the lanes that produced it
//   O^T[d][qrow]  += V^T[d][key] . P^T[key][qrow]    (V^T page rows as MFMA A operand)
// ------------------------------------------------------------------------------------
template <int HD, int HPW>
__global__ __launch_bounds__(256) void attn_chunk_kernel(const bf16_t *__restrict__ q, KvGeom kv, int layer, int nh, int G,
                                                         int64_t pos0, int n, int chunk, float scale,
                                                         float *__restrict__ part_o, float *__restrict__ part_ml) {
    constexpr int NKK = HD / 32, NDT = HD / 16;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int split = blockIdx.x, kvh = blockIdx.y;
    const int L = (int)(pos0 + n);
    const int c0 = split * chunk, c1 = min(L, c0 + chunk);
    const int head0 = kvh * G + w * HPW;
    const int qrow = lane & 15, qd = lane >> 4;

    frag_ab qf[HPW][NKK];
#pragma unroll
    for (int h = 0; h < HPW; ++h)
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            frag_ab z = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qrow < n) z = *reinterpret_cast<const frag_ab *>(q + (size_t)qrow * nh * HD + (size_t)(head0 + h) * HD + kk * 32 + qd * 8);
            qf[h][kk] = z;
        }
    f32x4 O[HPW][NDT];
    float mrun[HPW], lrun[HPW];
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        mrun[h] = -INFINITY;
        lrun[h] = 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) O[h][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int qpos = (int)pos0 + min(qrow, n - 1);
    const bf16_t *kbase = kv.k_pool + (size_t)layer * kv.layer_stride;
    const bf16_t *vbase = kv.vt_pool + (size_t)layer * kv.layer_stride;

    for (int kt0 = c0; kt0 < c1; kt0 += 32) {
        const int page = kv.page_table[kt0 / VLO_PAGE_TOKENS];
        const int tok0 = kt0 % VLO_PAGE_TOKENS;
        const bf16_t *kp = kbase + (size_t)page * kv.page_elems + ((size_t)kvh * VLO_PAGE_TOKENS + tok0) * HD;
        const bf16_t *vp = vbase + (size_t)page * kv.page_elems + ((size_t)kvh * HD) * VLO_PAGE_TOKENS + tok0;
        frag_ab kf[2][NKK];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk)
                kf[t][kk] = *reinterpret_cast<const frag_ab *>(kp + (size_t)(t * 16 + qrow) * HD + kk * 32 + qd * 8);
        frag_ab vf[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const bf16_t *vr = vp + (size_t)(dt * 16 + qrow) * VLO_PAGE_TOKENS + qd * 4;
            const uint2 lo = *reinterpret_cast<const uint2 *>(vr);
            const uint2 hi = *reinterpret_cast<const uint2 *>(vr + 16);
            const uint4 pk = make_uint4(lo.x, lo.y, hi.x, hi.y);
            vf[dt] = __builtin_bit_cast(frag_ab, pk);
        }
        const int kb = kt0 + qd * 4;
#pragma unroll
        for (int h = 0; h < HPW; ++h) {
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                s0 = mfma_bf16(kf[0][kk], qf[h][kk], s0);
                s1 = mfma_bf16(kf[1][kk], qf[h][kk], s1);
            }
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = (kb + r <= qpos) ? s0[r] * scale : -INFINITY;
                v[4 + r] = (kb + 16 + r <= qpos) ? s1[r] * scale : -INFINITY;
            }
            float tmax = v[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) tmax = fmaxf(tmax, v[j]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(mrun[h], tmax);
            const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __expf(mrun[h] - m_safe);
            mrun[h] = m_new;
            float psum = 0.f;
            frag_ab pb;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float p = __expf(v[j] - m_safe);
                psum += p;
                pb[j] = (short)f2bf(p);
            }
            lrun[h] = lrun[h] * alpha + psum;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                f32x4 o = O[h][dt];
                o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
                O[h][dt] = mfma_bf16(vf[dt], pb, o);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        float l = lrun[h];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const size_t row = ((size_t)split * nh + head0 + h) * 16 + qrow;
        if (qd == 0) {
            part_ml[row * 2] = mrun[h];
            part_ml[row * 2 + 1] = l;
        }
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const f32x4 o = O[h][dt];
            *reinterpret_cast<float4 *>(part_o + row * HD + dt * 16 + qd * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// grid = (nh, n); block = HD threads
__global__ void attn_combine_kernel(const float *__restrict__ part_o, const float *__restrict__ part_ml, int nsplit, int nh,
                                    int HD, bf16_t *__restrict__ out) {
    const int head = blockIdx.x, qrow = blockIdx.y, d = threadIdx.x;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part_ml[(((size_t)s * nh + head) * 16 + qrow) * 2]);
    float Lsum = 0.f, acc = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const size_t row = ((size_t)s * nh + head) * 16 + qrow;
        const float ms = part_ml[row * 2];
        const float wgt = (ms == -INFINITY) ? 0.f : __expf(ms - M);
        Lsum += part_ml[row * 2 + 1] * wgt;
        acc += part_o[row * HD + d] * wgt;
    }
    out[(size_t)qrow * nh * HD + (size_t)head * HD + d] = f2bf(acc / Lsum);
}

hipError_t attention_launch(const unsigned short *q, KvGeom kv, int layer, int num_heads, int64_t pos0, int n,
                            float *part_o, float *part_ml, unsigned short *out, hipStream_t st) {
    const int nkv = kv.num_kv_heads, hd = kv.head_dim, G = num_heads / nkv;
    const int L = (int)(pos0 + n);
    int target = (L + 127) / 128;
    const int want = (256 + nkv - 1) / nkv;           // ~one block per CU
    if (target > want) target = want;
    if (target > VLO_MAX_SPLITS) target = VLO_MAX_SPLITS;
    if (target < 1) target = 1;
    int chunk = (L + target - 1) / target;
    chunk = (chunk + 31) & ~31;
    const int nsplit = (L + chunk - 1) / chunk;
    const float scale = 1.0f / sqrtf((float)hd);
    int hpw = (G % 2 == 0) ? 2 : 1;
    if (G / hpw > 4) return hipErrorInvalidValue;     // block = (G/HPW) waves, launch bound 256 threads
    dim3 grid(nsplit, nkv), block((G / hpw) * 64);
#define VLO_ATTN(HD_, HPW_) \
    hipLaunchKernelGGL((attn_chunk_kernel<HD_, HPW_>), grid, block, 0, st, q, kv, layer, num_heads, G, pos0, n, chunk, scale, part_o, part_ml)
    if (hd == 128 && hpw == 2) VLO_ATTN(128, 2);
    else if (hd == 128 && hpw == 1) VLO_ATTN(128, 1);
    else if (hd == 64 && hpw == 2) VLO_ATTN(64, 2);
    else if (hd == 64 && hpw == 1) VLO_ATTN(64, 1);
    else return hipErrorInvalidValue;
#undef VLO_ATTN
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(attn_combine_kernel, dim3(num_heads, n), dim3(hd), 0, st, part_o, part_ml, nsplit, num_heads, hd, out);
    return hipGetLastError();
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
// samplers — one 1024-thread block over the L2-resident bf16 logits [V]
// ------------------------------------------------------------------------------------
struct ArgBest { float v; int i; };
VLO_DEV ArgBest better(ArgBest a, ArgBest b) {      // larger value wins; ties -> smaller index (torch argmax)
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}
VLO_DEV ArgBest block_argbest(ArgBest x, float *smv, int *smi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgBest y;
        y.v = __shfl_xor(x.v, o, 64);
        y.i = __shfl_xor(x.i, o, 64);
        x = better(x, y);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) { smv[w] = x.v; smi[w] = x.i; }
    __syncthreads();
    ArgBest r = {smv[0], smi[0]};
    for (int k = 1; k < nw; ++k) r = better(r, (ArgBest){smv[k], smi[k]});
    return r;
}

// force_mode: 0 = plain argmax; 1 = argmax but never eos (scheduled mode, mid-response);
//             2 = argmax computed, eos written (scheduled mode, last token)
__global__ __launch_bounds__(1024) void greedy_sample_kernel(const bf16_t *__restrict__ logits, int V, int64_t *tok_out, int eos,
                                                             int force_mode) {
    __shared__ float smv[16];
    __shared__ int smi[16];
    ArgBest b = {-INFINITY, 0x7fffffff};
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        const float v = bf2f(logits[i]);
        if (v > b.v) { b.v = v; b.i = i; }
    }
    b = block_argbest(b, smv, smi);
    if (threadIdx.x == 0) {
        int t = b.i;
        if (force_mode == 1 && t == eos) t = (eos + 1) % V;
        if (force_mode == 2) t = eos;
        *tok_out = t;
    }
}
hipError_t greedy_sample_launch(const unsigned short *logits, int V, int64_t *tok_out, int eos, int force_mode, hipStream_t st) {
    hipLaunchKernelGGL(greedy_sample_kernel, dim3(1), dim3(1024), 0, st, logits, V, tok_out, eos, force_mode);
    return hipGetLastError();
}

// demo/inference.py:76-79: softmax over a bf16 tensor (fp32 inside, bf16 out), threshold, argmax
__global__ __launch_bounds__(1024) void stream_sample_kernel(const bf16_t *__restrict__ logits, int V, float threshold,
                                                             int interval_id, int64_t *tok_out, float *p_interval_out) {
    __shared__ float sm[16];
    __shared__ float smv[16];
    __shared__ int smi[16];
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < V; i += blockDim.x) mx = fmaxf(mx, bf2f(logits[i]));
    mx = block_max(mx, sm);
    float s = 0.f;
    for (int i = threadIdx.x; i < V; i += blockDim.x) s += expf(bf2f(logits[i]) - mx);
    s = block_sum(s, sm);
    const float p_int = rbf(expf(bf2f(logits[interval_id]) - mx) / s);
    const bool zero_int = p_int < threshold;
    ArgBest b = {-INFINITY, 0x7fffffff};
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        float p = rbf(expf(bf2f(logits[i]) - mx) / s);
        if (i == interval_id && zero_int) p = 0.f;
        if (p > b.v) { b.v = p; b.i = i; }
    }
    b = block_argbest(b, smv, smi);
    if (threadIdx.x == 0) {
        *tok_out = b.i;
        if (p_interval_out) *p_interval_out = p_int;
    }
}
hipError_t stream_sample_launch(const unsigned short *logits, int V, float threshold, int interval_id, int64_t *tok_out,
                                float *p_interval_out, hipStream_t st) {
    if (interval_id < 0 || interval_id >= V) return hipErrorInvalidValue;
    hipLaunchKernelGGL(stream_sample_kernel, dim3(1), dim3(1024), 0, st, logits, V, threshold, interval_id, tok_out, p_interval_out);
    return hipGetLastError();
}
