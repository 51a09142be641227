// xcd_barrier_probe.hip — the price of a barrier among the 32 CUs of ONE XCD (DESIGN.md section 8.1 called it "the first thing to measure" before
// deciding on an LDS-resident per-layer ViT kernel), next to a flat chip-wide barrier over the same 256 workgroups.
//   XCD-local: arrive = one relaxed WORKGROUP-scope atomic add on the XCD's own counter line (atomics execute in the XCD's L2, which is coherent for
//   the XCD's CUs), poll = relaxed sc1 loads (bypass the CU's L1, served by the same L2); no release / acquire fence: nothing has to leave the XCD.
//   Hand-off check: before arriving every workgroup stores 4 KiB stamped with the round number (plain stores + s_waitcnt vmcnt(0)), after the barrier it
//   reads the 4 KiB of another workgroup OF ITS XCD with sc1 loads and counts words that do not carry the round's stamp (stale = protocol broken).
//   Flat: one agent-scope counter for all 256 workgroups, release fence before the arrive, acquire fence after the poll (the textbook form).
// Output: microseconds per barrier (wall over `rounds` back-to-back barriers) and the stale-word count.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/xcd_barrier_probe.hip -o tools/_bin/xcd_barrier_probe ; tools/_bin/xcd_barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Ctl {
    unsigned census[8][32];       // workgroups seen per XCD (one 128-B line each)
    unsigned slot[8][32];         // ticket dispenser per XCD: a workgroup's index inside its XCD
    unsigned ctr[8][32];          // XCD-local barrier counters
    unsigned flat[32];            // chip-wide barrier counter
    unsigned stale, timeouts;
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

// MODE 0: XCD-local barrier only; 1: + 4 KiB hand-off inside the XCD; 2: flat chip-wide barrier (release / acquire fences); 3: flat + hand-off to any workgroup;
// 4: as 1, but every round writes / reads a FRESH 4 KiB region (the reader's L1 has never seen those lines in this launch — the situation of
// csrc/vit_band.inc, where a buffer is written and read once per launch) and the reads are PLAIN loads
template <int MODE>
__global__ __launch_bounds__(256) void barrier_kernel(Ctl *c, unsigned *payload, int rounds, int per_xcd) {
    __shared__ unsigned s_slot, s_xcc, s_dead;
    if (threadIdx.x == 0) {
        s_dead = 0;
        const unsigned x = xcc_id();
        s_xcc = x;
        s_slot = __hip_atomic_fetch_add(&c->slot[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&c->census[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned x = s_xcc, me = s_slot;
    const bool flat = MODE == 2 || MODE == 3;
    constexpr int REGIONS = 64;                // MODE 4: payload regions per workgroup slot, one per round (rounds <= REGIONS)
    unsigned *mine = payload + ((size_t)(flat ? blockIdx.x : x * 64 + me)) * 1024;
    unsigned bad = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (MODE == 4) mine = payload + ((size_t)((r - 1) % REGIONS) * 512 + x * 64 + me) * 1024;
        if (MODE == 1 || MODE == 3 || MODE == 4) {
            reinterpret_cast<uint4 *>(mine)[threadIdx.x] = make_uint4(r, r, r, r);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (!flat) {
                __hip_atomic_fetch_add(&c->ctr[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned want = (unsigned)r * (unsigned)per_xcd;
                int spins = 0;
                while (__hip_atomic_load(&c->ctr[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    if (++spins > (1 << 21)) { atomicAdd(&c->timeouts, 1u); s_dead = 1; break; }
                }
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(&c->flat[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = (unsigned)r * gridDim.x;
                int spins = 0;
                while (__hip_atomic_load(&c->flat[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 21)) { atomicAdd(&c->timeouts, 1u); s_dead = 1; break; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        __syncthreads();
        if (s_dead) break;           // a bounded spin ran out (uneven placement): stop instead of spinning through every round
        if (MODE == 1 || MODE == 3 || MODE == 4) {
            const unsigned other = flat ? (blockIdx.x + 37u) % gridDim.x : x * 64 + (me + 5u) % (unsigned)per_xcd;
            const unsigned *src = payload + (MODE == 4 ? (size_t)((r - 1) % REGIONS) * 512 * 1024 : 0) + (size_t)other * 1024 + threadIdx.x * 4;
            uint4 v;
            if (!flat && MODE != 4) {       // sc1 loads: bypass this CU's L1, served by the XCD's L2
                v.x = __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v.y = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v.z = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v.w = __hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                v = *reinterpret_cast<const uint4 *>(src);
            }
            bad += (v.x < (unsigned)r) + (v.y < (unsigned)r) + (v.z < (unsigned)r) + (v.w < (unsigned)r);     // older than this round = stale (the writer may already be a round ahead)
        }
    }
    if (bad) atomicAdd(&c->stale, bad);
}

int main() {
    Ctl *c;
    unsigned *payload;
    CK(hipMalloc(&c, sizeof(Ctl)));
    CK(hipMalloc(&payload, (size_t)64 * 8 * 64 * 4096));          // 64 regions (MODE 4) of 8 x 64 slots of 4 KiB
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int rounds = 400, blocks = 256;
    // census first: how many workgroups of a 256-workgroup grid land on each XCD (the barrier needs the count)
    CK(hipMemset(c, 0, sizeof(Ctl)));
    hipLaunchKernelGGL(barrier_kernel<0>, dim3(blocks), dim3(256), 0, st, c, payload, 0, 32);
    CK(hipStreamSynchronize(st));
    Ctl h;
    CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
    printf("census of %d workgroups:", blocks);
    bool even = true;
    for (int x = 0; x < 8; ++x) { printf(" xcd%d=%u", x, h.census[x][0]); even = even && h.census[x][0] == (unsigned)blocks / 8; }
    printf("%s\n", even ? "" : "   (uneven: the XCD-local rows below are skipped)");
    const char *names[5] = {"XCD-local barrier (32 workgroups per XCD, 8 XCDs at once)", "XCD-local barrier + 4 KiB hand-off inside the XCD",
                            "flat chip-wide barrier, release + acquire fences", "flat chip-wide barrier + 4 KiB hand-off",
                            "XCD-local barrier + 4 KiB hand-off, fresh region per round, PLAIN loads"};
    for (int mode = 0; mode < 5; ++mode) {
        if ((mode < 2 || mode == 4) && !even) continue;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(c, 0, sizeof(Ctl)));
            CK(hipMemset(payload, 0, (size_t)64 * 8 * 64 * 4096));
            CK(hipEventRecord(e0, st));
            if (mode == 0) hipLaunchKernelGGL(barrier_kernel<0>, dim3(blocks), dim3(256), 0, st, c, payload, rounds, blocks / 8);
            if (mode == 1) hipLaunchKernelGGL(barrier_kernel<1>, dim3(blocks), dim3(256), 0, st, c, payload, rounds, blocks / 8);
            if (mode == 2) hipLaunchKernelGGL(barrier_kernel<2>, dim3(blocks), dim3(256), 0, st, c, payload, rounds, blocks / 8);
            if (mode == 3) hipLaunchKernelGGL(barrier_kernel<3>, dim3(blocks), dim3(256), 0, st, c, payload, rounds, blocks / 8);
            if (mode == 4) hipLaunchKernelGGL(barrier_kernel<4>, dim3(blocks), dim3(256), 0, st, c, payload, 64, blocks / 8);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
            if (rep == 1) printf("%-72s: %6.2f us per barrier   stale words %u   timeouts %u\n", names[mode], ms * 1e3 / (mode == 4 ? 64 : rounds), h.stale, h.timeouts);
        }
    }
    return 0;
}
