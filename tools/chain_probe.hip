// chain_probe.hip — feasibility probe for "chained launches": a decoder layer's chain of dependent weight-streaming kernels
// (1) on ONE stream (kernel boundaries are the dependencies, what engine.hip does today) versus
// (2) alternating between TWO streams, the dependency being a device-side completion counter: the consumer kernel is already
//     resident, has its first weight fragments in flight and waits on the counter before it touches the producer's output
//     (release / acquire idiom of the round-1 grid barrier: stores drained -> block barrier -> agent release fence -> relaxed
//     fetch_add; consumer: relaxed agent-scope poll -> agent acquire fence -> block barrier).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/chain_probe tools/chain_probe.hip ; run: tools/chain_probe [layers] [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned __attribute__((address_space(1))) gu32;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define NSUB 16
#define LINE 32            // unsigned words per 128-B line
#define PF 16              // uint4 in flight per thread (16 KiB per wave)

struct Dep { unsigned *ctr; unsigned want[NSUB]; };

__device__ __forceinline__ long long clk() { return (long long)__builtin_amdgcn_s_memrealtime(); }

__global__ __launch_bounds__(512) void stage(const v4u *__restrict__ W, int iters, const float *__restrict__ in, float *__restrict__ out,
                                             int n_io, Dep wait, unsigned *done, int chained, unsigned *err) {
    const int tid = threadIdx.x;
    const v4u *wp = W + (size_t)blockIdx.x * iters * 512 + tid;
    v4u r[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i)
        if (i < iters) r[i] = __builtin_nontemporal_load(wp + (size_t)i * 512);
    if (chained && wait.ctr) {
        if (tid < NSUB) {
            const long long t0 = clk();
            while ((int)(__hip_atomic_load((gu32 *)(wait.ctr + tid * LINE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - wait.want[tid]) < 0) {
                if (clk() - t0 > 100000000ll) { __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (tid < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
    }
    float acc = 0.f;
    const float x = in[tid % n_io];                 // the producer's output
    for (int i0 = 0; i0 < iters; i0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            if (i0 + i < iters) {
                const v4u v = r[i];
                acc += __uint_as_float((v.x ^ v.y ^ v.z ^ v.w) & 0x007fffffu) * 1e-30f * x;
                if (i0 + i + PF < iters) r[i] = __builtin_nontemporal_load(wp + (size_t)(i0 + i + PF) * 512);
            }
        }
    }
    // every block writes its slice of the output: out[j] = in[j] + 1 (+ 0 * acc keeps the loads alive)
    for (int j = blockIdx.x * 512 + tid; j < n_io; j += gridDim.x * 512) out[j] = in[j] + 1.0f + (acc != acc ? 1.f : 0.f);
    if (chained) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add((gu32 *)(done + (blockIdx.x % NSUB) * LINE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct Stage { const char *name; double mb; int blocks; };

int main(int argc, char **argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 32, reps = argc > 2 ? atoi(argv[2]) : 5;
    // one decoder layer of Llama-3-8B at n = 11, Lc ~ 6.6 k: bytes each phase streams
    const Stage st[7] = {{"add_rmsnorm", 0.5, 16}, {"qkv", 50.3, 256}, {"attention", 27.0, 256}, {"combine", 2.0, 64},
                         {"o_proj", 33.5, 256}, {"gate_up", 235.0, 256}, {"down", 117.0, 256}};
    const size_t pool_bytes = (size_t)3 << 30;
    v4u *W;
    CK(hipMalloc(&W, pool_bytes));
    CK(hipMemset(W, 0x11, pool_bytes));
    const int n_io = 11 * 4096;
    float *buf[2];
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&buf[i], n_io * 4)); CK(hipMemset(buf[i], 0, n_io * 4)); }
    unsigned *ctr, *err;
    const int RING = 8;
    CK(hipMalloc(&ctr, RING * NSUB * LINE * 4));
    CK(hipMalloc(&err, 4));
    hipStream_t s[2];
    CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    hipEvent_t e0, e1, ej;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    for (int mode = 0; mode < 3; ++mode) {       // 0: one stream; 1: two streams + counters; 2: one stream + counters (cost of the idiom alone)
        double best = 1e30;
        float final_v = 0;
        for (int rep = 0; rep < reps; ++rep) {
            CK(hipMemset(ctr, 0, RING * NSUB * LINE * 4));
            CK(hipMemset(err, 0, 4));
            CK(hipMemset(buf[0], 0, n_io * 4));
            CK(hipDeviceSynchronize());
            unsigned cum[RING][NSUB] = {};
            size_t woff = 0;
            int seq = 0;
            CK(hipEventRecord(e0, s[0]));
            if (mode == 1) { CK(hipEventRecord(ej, s[0])); CK(hipStreamWaitEvent(s[1], ej, 0)); }
            Dep prev{};
            for (int l = 0; l < layers; ++l)
                for (int k = 0; k < 7; ++k, ++seq) {
                    const int blocks = st[k].blocks;
                    int iters = (int)(st[k].mb * 1e6 / 16 / 512 / blocks);
                    if (iters < 1) iters = 1;
                    const size_t need = (size_t)blocks * iters * 512;
                    if (woff + need > pool_bytes / 16) woff = 0;
                    const int slot = seq % RING;
                    Dep mine{};
                    mine.ctr = ctr + (size_t)slot * NSUB * LINE;
                    for (int b = 0; b < blocks; ++b) cum[slot][b % NSUB]++;
                    for (int c = 0; c < NSUB; ++c) mine.want[c] = cum[slot][c];
                    hipStream_t stq = s[mode == 1 ? (seq & 1) : 0];
                    hipLaunchKernelGGL(stage, dim3(blocks), dim3(512), 0, stq, W + woff, iters, buf[seq & 1], buf[(seq + 1) & 1], n_io,
                                       prev, mine.ctr, mode != 0, err);
                    prev = mine;
                    woff += need;
                }
            if (mode == 1) { CK(hipEventRecord(ej, s[1])); CK(hipStreamWaitEvent(s[0], ej, 0)); }
            CK(hipEventRecord(e1, s[0]));
            CK(hipEventSynchronize(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            CK(hipMemcpy(&final_v, buf[(layers * 7) & 1], 4, hipMemcpyDeviceToHost));
            unsigned herr;
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            if (herr) printf("mode %d: a wait TIMED OUT\n", mode);
        }
        printf("mode %d (%s): %d layers x 7 kernels: %.3f ms  = %.1f us per layer; chain value %.0f (expect %d)\n", mode,
               mode == 0 ? "one stream, kernel boundaries" : mode == 1 ? "two streams, completion counters" : "one stream + counters",
               layers, best, best * 1e3 / layers, final_v, layers * 7);
    }
    return 0;
}
