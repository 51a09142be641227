// engine_probe.hip — go / no-go probe (round-3 verdict item 3b) for the CDNA guide's weight-streaming ENGINE on ONE dependency edge of
// the Llama-3-8B decoder layer: gate/up -> down.  SYNTHETIC: the right bytes move along the right paths with the right
// dependencies (HBM -> LDS ring by a run-ahead loader wave, LDS -> MFMA by three consumer waves, the activations crossing CUs
// as 8-byte data-tagged granules), the arithmetic is a stand-in.  What it decides: whether the pair as ONE persistent launch,
// with the loader running ahead ACROSS the edge (weights do not depend on activations), beats the same engine as two launches
// — and the two real GEMV launches (42.4 + 20.0 us at n = 11, profiles/r4_kernel_stats_bench200*.csv) — by the 5 % the verdict
// sets as the bar for building the real thing.
//
// Geometry (one workgroup per CU, 256 workgroups x 4 waves: wave 0 = loader, waves 1-3 = consumers):
//   phase 1 (gate/up, 235 MB): CU c streams its 7 column tiles of 16 rows x K = 4096  = 56 ring slots of 16 KiB and ends up
//           owning the 56 SwiGLU activation columns [56 c, 56 c + 56) of the n token rows;
//   edge:   a CU publishes its n x 56 bf16 activations as 8-byte {2 x bf16, epoch tag} granules (sc1 stores, one hop);
//   phase 2 (down, 117 MB): CU (i, j) = (c / 16, c % 16) takes K slice i (the 896 activation columns of the 16 CUs of its group) of
//           the 256 output columns j: 28 slots; it gathers the group's 16 x 308 granules (sc1 loads, 16 per lane per sweep,
//           re-swept until every tag is this launch's epoch) and writes a [n][256] fp32 partial (16 K slices: the reducing
//           row kernel reads 16 slabs instead of 4 — not modelled here, it is the next kernel's cost).
// LDS ring: 8 slots x 16 KiB; the loader issues a slot as 16 global_load_lds_dwordx4 (1 KiB each, nt) in inline asm (hipcc must
// not see them: it would drain the queue before every LDS flag read), keeps 3 fills in flight by counted vmcnt and publishes a
// landed slot through an LDS word; a consumer takes whole slots round-robin, reads 16 fragments (ds_read_b128) into 16 MFMAs and
// frees the slot through another LDS word.  Every spin is bounded (s_memrealtime) and raises `err`.
//
// modes: 0 = phase 1 alone, 1 = phase 2 alone (gather finds its granules published by the previous launch), 2 = both phases in one
// launch, loader runs ahead across the edge, 3 = both phases, loader held at the edge until the gather is done (isolates the
// prefetch credit).  Output: us per pair for (0 + 1 as two launches), 2, 3.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/engine_probe.hip -o tools/_bin/engine_probe ; run: tools/_bin/engine_probe [n_rows] [reps] [null_stream]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) short frag_ab;
typedef __attribute__((ext_vector_type(8))) __bf16 frag_bf;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned long long u64;
typedef u64 __attribute__((address_space(1))) gu64;

#define SLOT 16384
#define NSLOT 8
#define NCU 256
#define GROUP 16
#define SPIN_TICKS 4000000ll      // 40 ms of the 100 MHz counter: far beyond any legal wait, short enough not to hang the box

struct Args {
    const char *W1, *W2;        // this iteration's weights: phase 1 [NCU][slots1][SLOT], phase 2 [NCU][slots2][SLOT]
    int slots1, slots2;
    u64 *gran;                  // [NCU][gran_per_cu] granules
    int gran_per_cu;            // n * 56 / 2
    unsigned epoch;
    int mode;
    float *out;                 // [NCU][64 * 4] something that depends on everything (keeps the work alive)
    unsigned *err;
};

__device__ __forceinline__ long long clk() { return (long long)__builtin_amdgcn_s_memrealtime(); }

// one 1-KiB piece HBM -> LDS: lane l moves 16 bytes from gsrc (per lane) to lds_dst + 16 l (lds_dst wave-uniform); invisible to hipcc's
// s_waitcnt bookkeeping on purpose (cdna guide section 5.7)
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(256) void engine_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // the flag words are LDS-typed pointers on purpose: through a generic pointer hipcc emits flat_load / flat_store + s_waitcnt vmcnt(0),
    // which would drain the loader's direct-to-LDS queue at every flag access
    typedef __attribute__((address_space(3))) int lds_int;
    volatile lds_int *ready = (volatile lds_int *)(lds + NSLOT * SLOT);      // ready[s] = 1 + index of the item that has landed in slot s
    volatile lds_int *freed = ready + NSLOT;                                  // freed[s] = how many times slot s has been released
    volatile lds_int *flags = freed + NSLOT;                                  // [0] unused, [1] activations gathered, [2] abort, [4 + c] consumer c done with phase 1
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cu = blockIdx.x;
    if (tid < 2 * NSLOT + 8) ready[tid] = 0;
    __syncthreads();
    const bool p1 = a.mode != 1, p2 = a.mode != 0;
    const int n1 = p1 ? a.slots1 : 0, total = n1 + (p2 ? a.slots2 : 0);
    const unsigned lds_base = (unsigned)(size_t)lds;       // LDS byte address of the ring (generic -> local: low 32 bits)

#define SPIN_UNTIL(cond)                                                                         \
    do {                                                                                         \
        const long long t0_ = clk();                                                             \
        while (!(cond)) {                                                                        \
            if (flags[2]) break;                                                                 \
            if (clk() - t0_ > SPIN_TICKS) { flags[2] = 1; atomicOr(a.err, 1u << (wave & 3)); break; } \
            __builtin_amdgcn_s_sleep(2);                                                         \
        }                                                                                        \
    } while (0)

    if (wave == 0) {
        // ---------------- loader ----------------
        for (int i = 0; i < total; ++i) {
            const int s = i % NSLOT, use = i / NSLOT;
            SPIN_UNTIL(freed[s] >= use);
            if (a.mode == 3 && i == n1) {
                // held at the edge: everything issued so far has to land and be published first (the consumers need it to reach the edge)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) { if (i >= 2) ready[(i - 2) % NSLOT] = i - 1; if (i >= 1) ready[(i - 1) % NSLOT] = i; }
                SPIN_UNTIL(flags[1] != 0);
            }
            if (flags[2]) break;
            const char *src = (i < n1 ? a.W1 + ((size_t)cu * a.slots1 + i) * SLOT : a.W2 + ((size_t)cu * a.slots2 + (i - n1)) * SLOT) + lane * 16;
            const unsigned dst = lds_base + s * SLOT;
#pragma unroll
            for (int p = 0; p < 16; ++p) glds16(src + p * 1024, __builtin_amdgcn_readfirstlane(dst + p * 1024));
            if (i >= 2) {                                   // at most 2 fills (32 loads) still in flight: fill i - 2 has landed
                asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                if (lane == 0) ready[(i - 2) % NSLOT] = i - 1;
            }
        }
        if (total >= 2) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); if (lane == 0) ready[(total - 2) % NSLOT] = total - 1; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && total >= 1) ready[(total - 1) % NSLOT] = total;
        return;
    }
    // ---------------- consumers ----------------
    const int c = wave - 1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    frag_ab x;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (short)(0x3c00 + lane + j);
    bool edge_done = !(p1 && p2);
    // the group's granules: sweeps of 16 loads per lane (8 KB per sweep), each repeated until every tag carries this launch's epoch
    auto gather = [&]() {
        const int grp = cu / GROUP, ngr = GROUP * a.gran_per_cu;
        unsigned sum = 0;
        const long long t0 = clk();
        for (int k0 = 0; k0 < ngr; k0 += 64 * 16) {
            bool ok;
            do {
                u64 v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int idx = min(k0 + k * 64 + lane, ngr - 1);
                    v[k] = __hip_atomic_load((gu64 *)(a.gran + (size_t)(grp * GROUP + idx / a.gran_per_cu) * a.gran_per_cu + idx % a.gran_per_cu),
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                bool mine = true;
#pragma unroll
                for (int k = 0; k < 16; ++k) { mine = mine && (unsigned)(v[k] >> 32) == a.epoch; sum += (unsigned)v[k]; }
                ok = __builtin_amdgcn_ballot_w64(!mine) == 0;
                if (!ok && clk() - t0 > SPIN_TICKS) { flags[2] = 1; atomicOr(a.err, 16u); ok = true; }
                if (!ok) __builtin_amdgcn_s_sleep(1);
            } while (!ok);
            // staging into LDS (one ds_write_b32 per granule in a real engine): modelled by one write per sweep and lane
            ((volatile lds_int *)(lds + NSLOT * SLOT + 256))[lane] = (int)sum;
        }
        x[0] = (short)(x[0] + (short)(sum & 1));
        if (lane == 0) flags[1] = 1;
    };
    if (!p1 && p2 && c == 0) gather();      // phase 2 as its own launch: the previous launch published with this epoch, one sweep finds everything
    for (int i = c; i < total; i += 3) {
        if (i >= n1 && !edge_done) {
            // ---- the edge: this consumer has finished its phase-1 slots
            if (lane == 0) flags[4 + c] = 1;
            if (c == 0) {
                SPIN_UNTIL(flags[4] && flags[5] && flags[6]);
                // publish this CU's activations: n x 56 bf16 as granules {data, epoch}
                for (int g = lane; g < a.gran_per_cu; g += 64)
                    __hip_atomic_store((gu64 *)(a.gran + (size_t)cu * a.gran_per_cu + g), ((u64)a.epoch << 32) | (unsigned)(g + cu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gather();
            } else {
                SPIN_UNTIL(flags[1] != 0);
            }
            edge_done = true;
        }
        if (flags[2]) break;
        const int s = i % NSLOT;
        SPIN_UNTIL(ready[s] >= i + 1);
        if (flags[2]) break;
        const frag_ab *slot = reinterpret_cast<const frag_ab *>(lds + s * SLOT) + lane;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const frag_ab w = slot[p * 64];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(frag_bf, w), __builtin_bit_cast(frag_bf, x), acc, 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) freed[s] = i / NSLOT + 1;
    }
    if (p1 && !p2) {
        // phase 1 as its own launch: publish the activations for the next launch (plain kernel boundary = the hand-off)
        if (lane == 0) flags[4 + c] = 1;
        if (c == 0) {
            SPIN_UNTIL(flags[4] && flags[5] && flags[6]);
            for (int g = lane; g < a.gran_per_cu; g += 64)
                __hip_atomic_store((gu64 *)(a.gran + (size_t)cu * a.gran_per_cu + g), ((u64)a.epoch << 32) | (unsigned)(g + cu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // the phase's result (phase 2: the [n][256] fp32 partial of this CU): 4 floats per lane and consumer
    reinterpret_cast<f32x4 *>(a.out)[((size_t)cu * 3 + c) * 64 + lane] = acc;
}

int main(int argc, char **argv) {
    const int n_rows = argc > 1 ? atoi(argv[1]) : 11, reps = argc > 2 ? atoi(argv[2]) : 40;
    const int slots1 = 56, slots2 = 28, gran_per_cu = n_rows * 56 / 2;
    const size_t b1 = (size_t)NCU * slots1 * SLOT, b2 = (size_t)NCU * slots2 * SLOT;      // 235 MB, 117 MB
    const int nbuf = 8;                                                                  // 2.8 GB pool: a buffer is re-read after 2.5 GB of other traffic
    char *W;
    CK(hipMalloc(&W, (b1 + b2) * nbuf));
    CK(hipMemset(W, 0x3c, (b1 + b2) * nbuf));
    u64 *gran;
    CK(hipMalloc(&gran, (size_t)NCU * gran_per_cu * 8));
    CK(hipMemset(gran, 0, (size_t)NCU * gran_per_cu * 8));
    float *out;
    CK(hipMalloc(&out, (size_t)NCU * 3 * 64 * 16));
    unsigned *err;
    CK(hipMalloc(&err, 4));
    CK(hipMemset(err, 0, 4));
    const size_t lds = NSLOT * SLOT + 1024;
    CK(hipFuncSetAttribute((const void *)engine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t st = nullptr;                                       // argv[3] = 1: the legacy default stream (what bench_gemv / torch's default stream use)
    if (!(argc > 3 && atoi(argv[3]))) CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned epoch = 1;
    auto launch = [&](int mode, int it) {
        Args a{};
        a.W1 = W + (size_t)(it % nbuf) * (b1 + b2);
        a.W2 = a.W1 + b1;
        a.slots1 = slots1; a.slots2 = slots2; a.gran = gran; a.gran_per_cu = gran_per_cu; a.epoch = epoch; a.mode = mode; a.out = out; a.err = err;
        hipLaunchKernelGGL(engine_kernel, dim3(NCU), dim3(256), lds, st, a);
    };
    auto timed = [&](const char *name, auto body) {
        for (int it = 0; it < 3; ++it) { body(it); ++epoch; }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int it = 0; it < reps; ++it) { body(it + 3); ++epoch; }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h = 0;
        CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
        printf("%-64s %8.2f us per pair%s\n", name, ms * 1e3 / reps, h ? "   [SPIN TIMEOUT: err flags set, numbers invalid]" : "");
        if (h) { printf("err = 0x%x\n", h); exit(2); }
        return ms * 1e3 / reps;
    };
    printf("# engine_probe: gate/up (235 MB) -> down (117 MB) of a Llama-3-8B layer, n = %d rows, %d granules of 8 B per CU, %d reps\n", n_rows, gran_per_cu, reps);
    const double t_p1 = timed("phase 1 alone (one launch, no edge)", [&](int it) { launch(0, it); });
    const double t_two = timed("two launches (phase 1; phase 2 gathers what 1 published)", [&](int it) { launch(0, it); launch(1, it); });
    const double t_one = timed("ONE launch, loader runs ahead across the edge", [&](int it) { launch(2, it); });
    const double t_held = timed("one launch, loader HELD at the edge until the gather is done", [&](int it) { launch(3, it); });
    printf("# bytes / 6.3 TB/s = %.1f us;  one launch vs two: %.3fx;  prefetch credit (held - ahead): %.2f us;  phase 2 as a launch: %.2f us\n",
           (b1 + b2) / 6.3e12 * 1e6, t_one / t_two, t_held - t_one, t_two - t_p1);
    printf("# the real GEMV launches at n = 11 (rocprofv3, whole stream): gate/up 42.4 + down 20.0 = 62.4 us\n");
    return 0;
}
