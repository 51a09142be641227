// layer.hip — one decoder layer of the 16-row Llama step as ONE persistent launch (opt-in: VLO_PERSISTENT=1).
//
// Stage P0 of DESIGN.md §7 item 1: the seven phases of run_chunk's layer
//     add_rmsnorm -> qkv GEMV [RoPE + KV append] -> chunk attention -> combine -> o GEMV [residual, sum of squares]
//     -> gate/up GEMV [RMSNorm on load, SwiGLU] -> down GEMV [K-split partials]
// run inside one kernel whose resident blocks (one per CU) walk the SAME virtual grids with the SAME kernel bodies
// (gemv_body.inc, attn_body.inc, rmsnorm_body.inc are included textually here and in the stand-alone kernels), separated by
// grid barriers.  Results are bit-identical to the launch-per-phase pipeline by construction.  Stage P1 (LayerArgs.prefetch):
// a block waiting at the barrier in front of a GEMV phase already has its first weight fragments of that phase in flight,
// so the HBM stream runs through the phase change instead of restarting after it.
//
// Grid barrier (flat form): one monotonic counter; every block arrives once per barrier and waits for base + k * gridDim.x.  Memory
// hand-off follows cdna_hip_programming.md §6 Guideline 16: every wave drains its stores, block barrier, ONE lane does the
// agent-scope release fence (+ the asm wait the compiler may not drop), the relaxed arrive, a relaxed sc1 poll with
// s_sleep, ONE agent-scope acquire fence, block barrier, then plain loads.  The spin is bounded (s_memrealtime).
#include <stdlib.h>

#include "common.cuh"
#include "gemv.h"
#include "layer.h"
#include "llm_ops.h"

typedef __attribute__((address_space(1))) unsigned int lgu32;

// `prefetch`: what this block wants in flight while it waits (the next GEMV phase's first weight fragments).  Seven of the
// eight waves issue it right after their stores have drained; the leader wave only after its arrival is out, because its
// release / acquire sequence waits on vmcnt(0), which counts those loads too.
//
// Two forms (LayerArgs.barrier_kind):
//   flat  every block: agent release fence, arrive on ONE counter, poll it, agent acquire fence.  Correct whatever the
//         block -> XCD placement is; 256 write-backs and a 256-way fan-in per barrier.
//   xcd   hierarchical (MI355X_MICROARCH.md, price-list row barrier-xcd): blocks are grouped by the XCD they RUN on (read from
//         the hardware, HW_REG_XCC_ID — no placement pattern is assumed; group sizes come from a census taken before the first
//         barrier of the launch, which is always flat).  A block drains its stores into its XCD's L2 and arrives on the group's
//         counter; the LAST arriver of a group does the one release fence for that L2, arrives on the top counter, waits for
//         all groups, acquires, and publishes the group's generation word; the others poll that word and acquire.
struct BarrierCtx {
    unsigned *flat_counter, *err, *xcd;      // xcd: census[8] | group counters[8] | group generations[8] | top, one 128-B line each
    long long timeout_ticks;
    unsigned flat_target;                    // arrivals the NEXT flat barrier waits for
    int kind, nb;
    int k;                                   // xcd barriers done in this launch
    int group, group_size, groups;           // this block's XCD, its population, populated XCDs (valid after the census barrier)
};
#define VLO_XCD_LINE 32                      // unsigned words per 128-B line
VLO_DEV unsigned *xcd_census(const BarrierCtx &b, int g) { return b.xcd + (size_t)g * VLO_XCD_LINE; }
VLO_DEV unsigned *xcd_count(const BarrierCtx &b, int g) { return b.xcd + (size_t)(8 + g) * VLO_XCD_LINE; }
VLO_DEV unsigned *xcd_gen(const BarrierCtx &b, int g) { return b.xcd + (size_t)(16 + g) * VLO_XCD_LINE; }
VLO_DEV unsigned *xcd_top(const BarrierCtx &b) { return b.xcd + (size_t)24 * VLO_XCD_LINE; }
#define VLO_XCD_WORDS (25 * VLO_XCD_LINE)

VLO_DEV unsigned ld_relaxed(unsigned *p) { return __hip_atomic_load((lgu32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// wait until *p has reached `want` (wrap-safe); false on time-out
VLO_DEV bool spin_until(unsigned *p, unsigned want, unsigned *err, long long timeout_ticks) {
    const long long t0 = wall_clock64();
    while ((int)(ld_relaxed(p) - want) < 0) {
        if (wall_clock64() - t0 > timeout_ticks) {                 // a block that is not resident, or a dead peer: do not hang
            __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // pinned host word
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    return true;
}

template <class F>
VLO_DEV void grid_barrier(BarrierCtx &b, F prefetch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every wave: its global stores have left the CU (they are in L2)
    const bool leader_wave = threadIdx.x < 64;
    if (!leader_wave) prefetch();
    __syncthreads();
    const bool xcd = b.kind == 1 && b.k > 0;                      // the first barrier of a launch is flat: it completes the census
    bool waits_top = true;
    if (threadIdx.x == 0) {
        if (!xcd) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the write-back is complete before the arrival is visible
            __hip_atomic_fetch_add((lgu32 *)b.flat_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned old = __hip_atomic_fetch_add((lgu32 *)xcd_count(b, b.group), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            waits_top = old + 1u == (unsigned)b.k * (unsigned)b.group_size;      // the last arriver of this XCD in this round
            if (waits_top) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");               // ONE write-back for the XCD's L2
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add((lgu32 *)xcd_top(b), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (leader_wave) prefetch();
    if (threadIdx.x == 0) {
        if (!xcd) {
            spin_until(b.flat_counter, b.flat_target, b.err, b.timeout_ticks);
        } else if (waits_top) {
            spin_until(xcd_top(b), (unsigned)b.k * (unsigned)b.groups, b.err, b.timeout_ticks);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store((lgu32 *)xcd_gen(b, b.group), (unsigned)b.k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            spin_until(xcd_gen(b, b.group), (unsigned)b.k, b.err, b.timeout_ticks);
        }
        if (!(xcd && waits_top)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!xcd) b.flat_target += (unsigned)b.nb;
    if (b.kind == 1) {
        if (b.k == 0) {                                           // census complete: everybody reads the populations
            int groups = 0;
            for (int g = 0; g < 8; ++g) groups += ld_relaxed(xcd_census(b, g)) != 0u;
            b.groups = groups;
            b.group_size = (int)ld_relaxed(xcd_census(b, b.group));
        }
        b.k += 1;
    }
}

// HW_REG_XCC_ID (id 20), bits [3:0]: the XCD this wave runs on
VLO_DEV int hw_xcc_id() { return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u); }

VLO_DEV void barrier_init(BarrierCtx &b, const LayerArgs &L, int nb) {
    b.flat_counter = L.bar_counter; b.err = L.bar_err; b.xcd = L.bar_xcd; b.timeout_ticks = L.bar_timeout_ticks;
    b.flat_target = L.bar_base + (unsigned)nb; b.kind = L.barrier_kind; b.nb = nb; b.k = 0;
    b.group = 0; b.group_size = nb; b.groups = 1;
    if (b.kind == 1) {
        b.group = hw_xcc_id();
        if (threadIdx.x == 0) __hip_atomic_fetch_add((lgu32 *)xcd_census(b, b.group), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The phase bodies derive their lane / wave / row indices from threadIdx.x; left alone, the compiler shares those address
// computations across the phases and keeps them in registers through the attention phase, which has none to spare (24 instead
// of 68 B of scratch per lane in the 8B kernel).  The attention body therefore sees its own opaque copy of the thread index.
struct TidX { unsigned x; };
VLO_DEV TidX opaque_tid() {
    unsigned v = threadIdx.x;
    asm volatile("" : "+v"(v));
    return TidX{v};
}

// ---- the phase bodies, from the same text as the stand-alone kernels ------------------------------------------------------
// (rounding points follow the reference's bf16 CPU path — see gemv.hip / llm_ops.hip)
VLO_DEV float silu_bf16(float g) { return rbf(g / (1.0f + __expf(-g))); }
VLO_DEV float gelu_python_bf16(float x) {
    const float a = rbf(x * 0.5f);
    const float t = rbf(x / 1.4142135623730951f);
    const float e = rbf(erff(t));
    const float s = rbf(1.0f + e);
    return rbf(a * s);
}
VLO_DEV float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// the first weight fragments of virtual block (vbx, vby) into `pre`: the geometry text of the GEMV body, nothing else
template <int KF, int NW, int EPI>
VLO_DEV void gemv16_preload(const GemvArgs &a, const int vbx, const int vby, float4 *red, frag_ab (&pre)[KF]) {
#define VLO_GEMV_BX vbx
#define VLO_GEMV_BY vby
#define VLO_GEMV_WR_REF pre
#include "gemv_head.inc"
#undef VLO_GEMV_BX
#undef VLO_GEMV_BY
#undef VLO_GEMV_WR_REF
    (void)rs_lds; (void)tmp_lds; (void)m16; (void)qd; (void)tile_b;
}

// the GEMV body for virtual block (vbx, vby) of a (vgx, *) grid whose first fragments are already in `pre`; `pre` IS the body's
// weight register set (it is reloaded by the rolling prefetch and holds nothing useful afterwards)
template <int KF, int NW, int XSRC, int EPI>
VLO_DEV void gemv16_dev(const GemvArgs &a, const int vbx, const int vby, const int vgx, float4 *red, frag_ab (&pre)[KF]) {
#define VLO_GEMV_BX vbx
#define VLO_GEMV_BY vby
#define VLO_GEMV_GX vgx
#define VLO_GEMV_WR_REF pre
#define VLO_GEMV_PRELOADED 1
#include "gemv_body.inc"
#undef VLO_GEMV_BX
#undef VLO_GEMV_BY
#undef VLO_GEMV_GX
#undef VLO_GEMV_WR_REF
#undef VLO_GEMV_PRELOADED
}

template <int HD, int HPW>
VLO_DEV void attn_chunk_dev(const bf16_t *q, KvGeom kv, int layer, int nh, int G, int KS, int64_t pos0, int n, int chunk, float scale,
                            float *part_o, float *part_ml, const int vbx, const int vby, const int vgx, float4 *lds_o) {
    const TidX vlo_tid = opaque_tid();
#define threadIdx vlo_tid
    do {                                                           // VLO_ATTN_EXIT leaves the body, not the kernel
#define VLO_ATTN_BX vbx
#define VLO_ATTN_BY vby
#define VLO_ATTN_BZ 0
#define VLO_ATTN_GX vgx
#define VLO_ATTN_EXIT break
#define VLO_ATTN_NO_KPREFETCH 1
#include "attn_body.inc"
#undef VLO_ATTN_NO_KPREFETCH
#undef VLO_ATTN_BX
#undef VLO_ATTN_BY
#undef VLO_ATTN_BZ
#undef VLO_ATTN_GX
#undef VLO_ATTN_EXIT
    } while (0);
#undef threadIdx
}

#define RMS_THREADS 512
#define RMS_MAXCH 2
VLO_DEV void add_rmsnorm_dev(bf16_t *h, const float *partial, int ksplit, int partial_ld, const bf16_t *w, bf16_t *x, int H, int ldx,
                             float eps, const int row) {
#define VLO_RMS_ROW row
#include "rmsnorm_body.inc"
#undef VLO_RMS_ROW
}

// attn_combine_kernel's arithmetic (llm_ops.hip) for TWO (head, query row) items at a time: threads 0..255 take item `it0`,
// threads 256..511 item `it0 + 1`; the loop trip count is uniform over the block (the barriers inside are block-wide)
VLO_DEV void combine_dev(const float *part_o, const float *part_ml, int nsplit, int nh, int HD, int m, bf16_t *out, int bid, int nblocks) {
    __shared__ float wgt[2][VLO_MAX_SPLITS];
    __shared__ float red2[2][256];
    __shared__ float Ltot[2];
    const int half = threadIdx.x >> 8, t = threadIdx.x & 255;
    const int items = nh * m;
    for (int it0 = bid * 2; it0 < items; it0 += nblocks * 2) {
        const int it = it0 + half;
        const bool live = it < items;
        const int head = live ? it % nh : 0, qrow = live ? it / nh : 0;
        if (t < 64) {                                  // one wave per half: softmax weights of the splits
            float ms = -INFINITY, ls = 0.f;
            if (live && t < nsplit) {
                const size_t row = ((size_t)t * nh + head) * 16 + qrow;
                ms = part_ml[row * 2];
                ls = part_ml[row * 2 + 1];
            }
            const float M = wave_max(ms);
            const float wv = (ms == -INFINITY) ? 0.f : __expf(ms - M);
            const float Ls = wave_sum(ls * wv);
            if (t < nsplit) wgt[half][t] = wv;
            if (t == 0) Ltot[half] = Ls;
        }
        __syncthreads();
        const int d = t % HD, ql = t / HD, nql = 256 / HD;
        float acc = 0.f;
        if (live)
            for (int s = ql; s < nsplit; s += nql) acc += part_o[(((size_t)s * nh + head) * 16 + qrow) * HD + d] * wgt[half][s];
        red2[half][t] = acc;
        __syncthreads();
        if (live && ql == 0) {
            for (int k2 = 1; k2 < nql; ++k2) acc += red2[half][k2 * HD + d];
            out[(size_t)qrow * nh * HD + (size_t)head * HD + d] = f2bf(acc / Ltot[half]);
        }
        __syncthreads();                               // wgt / red2 / Ltot are rewritten by the next pair of items
    }
}

// The seven phases of one decoder layer for the resident block `bid` of `nb`.  `bar`: the launch's barrier state; preH / preI / have carry prefetched weight fragments from one barrier to the phase behind it — and,
// in the whole-step kernel, from the end of one layer to the qkv phase of the next.  `next`: the following layer of the same
// launch (null in the per-layer kernel and after the last layer): its qkv fragments go in flight behind this layer's down-proj.
template <int KFH, int KFI, int HD, int HPW>
VLO_DEV void layer_phases(const LayerArgs &L, const LayerArgs *next, const int bid, const int nb, BarrierCtx &bar, float4 *lds,
                          frag_ab (&preH)[KFH], frag_ab (&preI)[KFI], bool &have) {
    auto nothing = []() {};
#define VLO_BARRIER(...) grid_barrier(bar, __VA_ARGS__)
#define VLO_GEMV_PHASE(KF_, XSRC_, EPI_, ARGS_, GX_, GY_, PRE_)                                                          \
    for (int vb = bid; vb < (GX_) * (GY_); vb += nb) {                                                                    \
        if (!(have && vb == bid)) gemv16_preload<KF_, 8, EPI_>(ARGS_, vb % (GX_), vb / (GX_), lds, PRE_);                 \
        gemv16_dev<KF_, 8, XSRC_, EPI_>(ARGS_, vb % (GX_), vb / (GX_), GX_, lds, PRE_);                                   \
        __syncthreads();                                                                                                  \
    }                                                                                                                     \
    have = false
    // phase 0: h (+)= bf16(sum of the previous layer's down-proj slices); x = RMSNorm(h) * w_in
    for (int vb = bid; vb < L.m; vb += nb) {
        add_rmsnorm_dev(L.h, L.prev, L.prev_ks, L.H, L.ln_in, L.x, L.H, L.H, L.eps, vb);
        __syncthreads();
    }
    VLO_BARRIER([&]() {      // (already in flight when the previous layer of this launch ended with the prefetch)
        if (L.prefetch && !have && bid < L.qkv_gx) { gemv16_preload<KFH, 8, EPI_ROPE>(L.qkv, bid, 0, lds, preH); have = true; }
    });
    // phase 1: q, k, v = x W^T; RoPE; K / V^T appended to the session's pages
    VLO_GEMV_PHASE(KFH, XSRC_PLAIN, EPI_ROPE, L.qkv, L.qkv_gx, 1, preH);
    VLO_BARRIER(nothing);
    // phase 2: split-KV attention partials, virtual grid (nsplit, kv heads)
    {
        const int items = L.nsplit * L.kv.num_kv_heads;
        for (int vb = bid; vb < items; vb += nb) {
            // the attention body is written for (head groups x key sub-splits) waves; surplus waves of this block only keep
            // the body's one block barrier (the key sub-split merge) company
            if ((int)threadIdx.x < L.attn_threads)
                attn_chunk_dev<HD, HPW>(L.q, L.kv, L.layer, L.nh, L.G, L.KS, L.pos0, L.m, L.chunk, L.scale, L.part_o, L.part_ml,
                                        vb % L.nsplit, vb / L.nsplit, L.nsplit, lds);
            else if (L.KS > 1)
                __syncthreads();
            __syncthreads();
        }
    }
    VLO_BARRIER([&]() {      // o_proj's fragments stay in registers across the (light) combine phase
        if (L.prefetch && bid < L.o_gx) { gemv16_preload<KFH, 8, EPI_RESID>(L.o, bid, 0, lds, preH); have = true; }
    });
    // phase 3: merge the splits
    combine_dev(L.part_o, L.part_ml, L.nsplit, L.nh, HD, L.m, L.attn_out, bid, nb);
    VLO_BARRIER(nothing);
    // phase 4: h += bf16(attn W_o^T), row sums of squares of the new h
    VLO_GEMV_PHASE(KFH, XSRC_PLAIN, EPI_RESID, L.o, L.o_gx, 1, preH);
    VLO_BARRIER([&]() {
        if (L.prefetch && bid < L.gu_gx) { gemv16_preload<KFH, 8, EPI_SWIGLU>(L.gu, bid, 0, lds, preH); have = true; }
    });
    // phase 5: act = silu(gate) * up, post-attention RMSNorm on the operand load
    VLO_GEMV_PHASE(KFH, XSRC_NORM, EPI_SWIGLU, L.gu, L.gu_gx, 1, preH);
    VLO_BARRIER([&]() {
        if (L.prefetch && bid < L.down_gx * L.down_gy) {
            gemv16_preload<KFI, 8, EPI_PARTIAL_F32>(L.down, bid % L.down_gx, bid / L.down_gx, lds, preI);
            have = true;
        }
    });
    // phase 6: down-proj K-slice partials (combined by the next layer's phase 0, or by add_rmsnorm after the last layer)
    VLO_GEMV_PHASE(KFI, XSRC_PLAIN, EPI_PARTIAL_F32, L.down, L.down_gx, L.down_gy, preI);
    if (next) {              // whole-step kernel: the boundary between two layers is one more barrier, not a launch
        VLO_BARRIER([&]() {
            if (next->prefetch && bid < next->qkv_gx) { gemv16_preload<KFH, 8, EPI_ROPE>(next->qkv, bid, 0, lds, preH); have = true; }
        });
    }
#undef VLO_GEMV_PHASE
#undef VLO_BARRIER
}

template <int KFH, int KFI, int HD, int HPW>
__global__ __launch_bounds__(512) void llm_layer_kernel(LayerArgs L) {
    extern __shared__ __attribute__((aligned(16))) float4 lds[];
    BarrierCtx bar;
    barrier_init(bar, L, gridDim.x);
    frag_ab preH[KFH], preI[KFI];
    bool have = false;
    layer_phases<KFH, KFI, HD, HPW>(L, nullptr, blockIdx.x, gridDim.x, bar, lds, preH, preI, have);
}

// all decoder layers of a step in ONE launch: `layers` = device array of the per-layer arguments (bar_base of the first one counts)
template <int KFH, int KFI, int HD, int HPW>
__global__ __launch_bounds__(512) void llm_step_kernel(const LayerArgs *__restrict__ layers, int num_layers) {
    extern __shared__ __attribute__((aligned(16))) float4 lds[];
    BarrierCtx bar;
    barrier_init(bar, layers[0], gridDim.x);
    frag_ab preH[KFH], preI[KFI];
    bool have = false;
    for (int l = 0; l < num_layers; ++l)
        layer_phases<KFH, KFI, HD, HPW>(layers[l], l + 1 < num_layers ? layers + l + 1 : nullptr, blockIdx.x, gridDim.x, bar, lds, preH, preI,
                                        have);
}

// ---- host side --------------------------------------------------------------------------------------------------------------
// FLAT-counter arrivals per block and launch (what the host adds to bar_base): all barriers of the launch with the flat form,
// only the first (census) one with the XCD-hierarchical form
int layer_barriers_per_launch(int barrier_kind) { return barrier_kind == 1 ? 1 : 6; }
int step_barriers_per_launch(int num_layers, int barrier_kind) { return barrier_kind == 1 ? 1 : 7 * num_layers - 1; }
int layer_xcd_words(void) { return VLO_XCD_WORDS; }

// cooperative launch: the runtime checks the grid against the occupancy query (a block that is not resident would leave the
// barriers waiting for their time-out).  VLO_PERSISTENT_COOP=0: a plain launch — same residency, no check, ~15 us less host time
static bool use_coop() {
    static const bool coop = getenv("VLO_PERSISTENT_COOP") ? atoi(getenv("VLO_PERSISTENT_COOP")) != 0 : true;
    return coop;
}

template <int KFH, int KFI, int HD, int HPW>
static hipError_t launch_one(LayerArgs &L, int nblocks, size_t lds, hipStream_t st) {
    static size_t granted = 0;
    if (!granted) {
        // opt in to dynamic LDS beyond the 64 KiB default: the largest size the runtime accepts next to the kernel's static LDS;
        // a refusal must not linger as the runtime's "last error" (the first hardware run failed on exactly that)
        granted = 64 * 1024;
        for (size_t want = 160 * 1024; want > 64 * 1024; want -= 8 * 1024) {
            if (hipFuncSetAttribute((const void *)llm_layer_kernel<KFH, KFI, HD, HPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) == hipSuccess) {
                granted = want;
                break;
            }
            (void)hipGetLastError();
        }
    }
    if (lds > granted) return hipErrorInvalidValue;
    if (!use_coop()) {
        hipLaunchKernelGGL((llm_layer_kernel<KFH, KFI, HD, HPW>), dim3(nblocks), dim3(512), lds, st, L);
        return hipGetLastError();
    }
    void *params[] = {&L};
    return hipLaunchCooperativeKernel(llm_layer_kernel<KFH, KFI, HD, HPW>, dim3(nblocks), dim3(512), params, (unsigned)lds, st);
}

template <int KFH, int KFI, int HD, int HPW>
static hipError_t launch_step(const LayerArgs *layers_dev, int num_layers, int nblocks, size_t lds, hipStream_t st) {
    static size_t granted = 0;
    if (!granted) {
        // opt in to dynamic LDS beyond the 64 KiB default: the largest size the runtime accepts next to the kernel's static LDS;
        // a refusal must not linger as the runtime's "last error" (the first hardware run failed on exactly that)
        granted = 64 * 1024;
        for (size_t want = 160 * 1024; want > 64 * 1024; want -= 8 * 1024) {
            if (hipFuncSetAttribute((const void *)llm_step_kernel<KFH, KFI, HD, HPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) == hipSuccess) {
                granted = want;
                break;
            }
            (void)hipGetLastError();
        }
    }
    if (lds > granted) return hipErrorInvalidValue;
    if (!use_coop()) {
        hipLaunchKernelGGL((llm_step_kernel<KFH, KFI, HD, HPW>), dim3(nblocks), dim3(512), lds, st, layers_dev, num_layers);
        return hipGetLastError();
    }
    void *params[] = {&layers_dev, &num_layers};
    return hipLaunchCooperativeKernel(llm_step_kernel<KFH, KFI, HD, HPW>, dim3(nblocks), dim3(512), params, (unsigned)lds, st);
}

bool layer_kernel_supports(int kf_h, int kf_i, int head_dim, int hpw) {
    return (kf_h == 16 && kf_i == 14 && head_dim == 128 && hpw == 2) ||      // Llama-3-8B
           (kf_h == 8 && kf_i == 11 && head_dim == 64 && hpw == 2) ||        // TinyLlama-1.1B
           (kf_h == 1 && kf_i == 1 && head_dim == 64 && hpw == 2);           // 256 / 768-wide model of the unit tests (down-proj in 3 K slices)
}

hipError_t layer_launch(LayerArgs &L, int kf_h, int kf_i, int head_dim, int hpw, int nblocks, size_t lds, hipStream_t st) {
    if (kf_h == 16 && kf_i == 14 && head_dim == 128 && hpw == 2) return launch_one<16, 14, 128, 2>(L, nblocks, lds, st);
    if (kf_h == 8 && kf_i == 11 && head_dim == 64 && hpw == 2) return launch_one<8, 11, 64, 2>(L, nblocks, lds, st);
    if (kf_h == 1 && kf_i == 1 && head_dim == 64 && hpw == 2) return launch_one<1, 1, 64, 2>(L, nblocks, lds, st);
    return hipErrorInvalidValue;
}

hipError_t step_launch(const LayerArgs *layers_dev, int num_layers, int kf_h, int kf_i, int head_dim, int hpw, int nblocks, size_t lds,
                       hipStream_t st) {
    if (kf_h == 16 && kf_i == 14 && head_dim == 128 && hpw == 2) return launch_step<16, 14, 128, 2>(layers_dev, num_layers, nblocks, lds, st);
    if (kf_h == 8 && kf_i == 11 && head_dim == 64 && hpw == 2) return launch_step<8, 11, 64, 2>(layers_dev, num_layers, nblocks, lds, st);
    if (kf_h == 1 && kf_i == 1 && head_dim == 64 && hpw == 2) return launch_step<1, 1, 64, 2>(layers_dev, num_layers, nblocks, lds, st);
    return hipErrorInvalidValue;
}
