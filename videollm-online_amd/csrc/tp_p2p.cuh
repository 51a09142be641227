// tp_p2p.cuh — device side of the one-shot peer-to-peer tensor-parallel exchange (host side and the protocol description:
// tp.hip "p2p exchange"; C ABI: include/vlo.h vlo_tp_p2p_*).  Kept in a header of its own: the unit tests compile these two
// kernels on their own.
#pragma once
#include "common.cuh"

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

struct P2PPeers { unsigned long long *mbox[8]; };

VLO_DEV void granule_store(unsigned long long *p, unsigned epoch, unsigned value) {
    __hip_atomic_store((gu64 *)p, ((unsigned long long)epoch << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
VLO_DEV unsigned long long granule_load(const unsigned long long *p) {
    return __hip_atomic_load((gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
VLO_DEV void p2p_timeout(unsigned *err_dev, unsigned *err_host) {
    __hip_atomic_fetch_or((gu32 *)err_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_or(err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct XchgArgs {
    const float *partial;        // this rank's partial sums [ks][16][ld] (fp32), produced by the preceding GEMV
    int ks, ld;
    P2PPeers peers;              // every rank's mailbox, by global rank
    int T, me;
    unsigned long long slot_off; // granule offset of [slot][0][0][0] in the reduce region
    unsigned epoch;
    int mode;                    // bit 0: publish, bit 1: collect + residual add + RMSNorm
    unsigned short *h;           // residual stream [16][H] bf16, updated in place
    const unsigned short *w;     // norm weight
    unsigned short *x;           // normed rows out [16][ldx]
    int H, ldx;
    float eps;
    unsigned *err_dev, *err_host;
    long long timeout_ticks;
};

#ifndef XCHG_THREADS
#define XCHG_THREADS 512
#endif
// grid = m rows; thread t owns the 8-column chunks t, t + 512, ... of its row (one chunk when H = 4096)
__global__ __launch_bounds__(XCHG_THREADS) void tp_xchg_norm_kernel(XchgArgs a) {
    __shared__ float sm[16];
    const int m = blockIdx.x;
    const int nch = a.H >> 3;
    const size_t row_off = a.slot_off + (size_t)m * a.H;
    const size_t src_stride = (size_t)16 * a.H;              // granules between two sources of one slot
    if (a.mode & 1) {
        for (int ch = threadIdx.x; ch < nch; ch += XCHG_THREADS) {
            float d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int s = 0; s < a.ks; ++s) {
                const float4 *pp = reinterpret_cast<const float4 *>(a.partial + ((size_t)s * 16 + m) * a.ld + ch * 8);
                const float4 u = pp[0], v = pp[1];
                d[0] += u.x; d[1] += u.y; d[2] += u.z; d[3] += u.w;
                d[4] += v.x; d[5] += v.y; d[6] += v.z; d[7] += v.w;
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                if (p < a.T) {
                    unsigned long long *dst = a.peers.mbox[p] + row_off + (size_t)a.me * src_stride + ch * 8;
#pragma unroll
                    for (int j = 0; j < 8; ++j) granule_store(dst + j, a.epoch, __float_as_uint(d[j]));
                }
            }
        }
    }
    if (!(a.mode & 2)) return;
    const unsigned long long *own = a.peers.mbox[a.me] + row_off;
    const bool dead = __hip_atomic_load((gu32 *)a.err_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    const unsigned all = (1u << a.T) - 1u;
    const long long t0 = wall_clock64();
    bf16_t *hr = a.h + (size_t)m * a.H;
    float ss = 0.f;
    // pass 1: collect, add to the residual stream, accumulate the row's sum of squares
    for (int ch = threadIdx.x; ch < nch; ch += XCHG_THREADS) {
        unsigned long long g[8][8];
        unsigned done = dead ? all : 0u;
        if (dead) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int j = 0; j < 8; ++j) g[p][j] = 0ull;
        }
        while (done != all) {
            // every pending source's 8 granules go in flight together, then the tags are checked
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if (p < a.T && !((done >> p) & 1u)) {
                    const unsigned long long *src = own + (size_t)p * src_stride + ch * 8;
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[p][j] = granule_load(src + j);
                }
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if (p < a.T && !((done >> p) & 1u)) {
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ok &= (unsigned)(g[p][j] >> 32) == a.epoch;
                    if (ok) done |= 1u << p;
                }
            if (done == all) break;
            if (wall_clock64() - t0 > a.timeout_ticks) {
                p2p_timeout(a.err_dev, a.err_host);
#pragma unroll
                for (int p = 0; p < 8; ++p)
                    if (!((done >> p) & 1u))
#pragma unroll
                        for (int j = 0; j < 8; ++j) g[p][j] = 0ull;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        float d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int p = 0; p < 8; ++p)                 // rank order: the same fp32 sum on every rank
            if (p < a.T)
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] += __uint_as_float((unsigned)g[p][j]);
        const uint4 raw = *reinterpret_cast<const uint4 *>(hr + ch * 8);
        const bf16_t *e = reinterpret_cast<const bf16_t *>(&raw);
        uint4 o;
        bf16_t *oe = reinterpret_cast<bf16_t *>(&o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {               // linear output -> bf16, then the bf16 residual add (as add_rmsnorm_kernel)
            const float hn = rbf(bf2f(e[j]) + rbf(d[j]));
            oe[j] = f2bf(hn);
            ss += hn * hn;
        }
        *reinterpret_cast<uint4 *>(hr + ch * 8) = o;
    }
    ss = block_sum(ss, sm);
    const float rs = 1.0f / sqrtf(ss / (float)a.H + a.eps);
    // pass 2: x = w * bf16(h * rs); every thread re-reads the chunks it wrote itself
    bf16_t *xr = a.x + (size_t)m * a.ldx;
    for (int ch = threadIdx.x; ch < nch; ch += XCHG_THREADS) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(hr + ch * 8);
        const uint4 wraw = *reinterpret_cast<const uint4 *>(a.w + ch * 8);
        const bf16_t *e = reinterpret_cast<const bf16_t *>(&raw), *we = reinterpret_cast<const bf16_t *>(&wraw);
        uint4 o;
        bf16_t *oe = reinterpret_cast<bf16_t *>(&o);
#pragma unroll
        for (int j = 0; j < 8; ++j) oe[j] = f2bf(bf2f(we[j]) * rbf(bf2f(e[j]) * rs));
        *reinterpret_cast<uint4 *>(xr + ch * 8) = o;
    }
}

struct GatherArgs {
    const unsigned short *local;   // this rank's logits shard [nr][Vl] bf16
    unsigned short *out;           // [nr][V] bf16
    P2PPeers peers;
    int T, me, Vl, V;
    unsigned long long slot_off;   // granule offset of [slot][0][0][0] in the gather region
    unsigned epoch;
    int mode;                      // bit 0: publish, bit 1: collect
    unsigned *err_dev, *err_host;
    long long timeout_ticks;
};
// grid = (blocks, nr rows); one granule = two adjacent bf16 logits
__global__ __launch_bounds__(256) void tp_gather_kernel(GatherArgs a) {
    const int row = blockIdx.y;
    const int Vh = a.Vl >> 1;
    const size_t src_stride = (size_t)16 * Vh;
    const size_t row_off = a.slot_off + (size_t)row * Vh;
    if (a.mode & 1) {
        const unsigned *loc = reinterpret_cast<const unsigned *>(a.local + (size_t)row * a.Vl);
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Vh; i += gridDim.x * blockDim.x) {
            const unsigned v = loc[i];
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if (p < a.T) granule_store(a.peers.mbox[p] + row_off + (size_t)a.me * src_stride + i, a.epoch, v);
        }
    }
    if (!(a.mode & 2)) return;
    const unsigned long long *own = a.peers.mbox[a.me] + row_off;
    const bool dead = __hip_atomic_load((gu32 *)a.err_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    const long long t0 = wall_clock64();
    const int total = a.T * Vh;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int p = i / Vh, c = i - p * Vh;
        unsigned long long g = 0ull;
        if (!dead) {
            for (;;) {
                g = granule_load(own + (size_t)p * src_stride + c);
                if ((unsigned)(g >> 32) == a.epoch) break;
                if (wall_clock64() - t0 > a.timeout_ticks) {
                    p2p_timeout(a.err_dev, a.err_host);
                    g = 0ull;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        reinterpret_cast<unsigned *>(a.out + (size_t)row * a.V + (size_t)p * a.Vl)[c] = (unsigned)g;
    }
}
