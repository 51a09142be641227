// launch_cost_probe.hip — what a weight-streaming kernel of the Llama step costs on this chip as a function of the bytes it reads: a chain of
// DEPENDENT launches on one stream (each reads `bytes` with 256 workgroups x 8 waves, 16 KiB per wave in flight, nontemporal 16-byte loads — the
// GEMV's access pattern without its arithmetic — and writes one word the next launch reads), timed with HIP events over the chain.
// DESIGN.md section 7 reads every small kernel of the step as "fixed cost + bytes / achievable rate"; this probe measures both terms directly.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/launch_cost_probe tools/launch_cost_probe.hip ;  tools/_bin/launch_cost_probe [launches]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define PF 16              // 16-byte loads in flight per thread (16 KiB per wave)

__global__ __launch_bounds__(512) void stream_kernel(const v4u *__restrict__ W, long iters, const unsigned *__restrict__ in, unsigned *__restrict__ out) {
    const unsigned dep = in[0];                                  // the previous launch's word: a true dependency
    const v4u *wp = W + (size_t)blockIdx.x * iters * 512 + threadIdx.x;
    v4u acc = {dep, 0u, 0u, 0u};
    v4u r[PF];
    // iters is a multiple of PF (host): no load of the loop sits behind a predicate, so hipcc's waits are counted (rolling vmcnt(15 .. 0))
#pragma unroll
    for (int i = 0; i < PF; ++i) r[i] = __builtin_nontemporal_load(wp + (size_t)i * 512);
    for (long i0 = PF; i0 < iters; i0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            acc ^= r[i];
            r[i] = __builtin_nontemporal_load(wp + (size_t)(i0 + i) * 512);
            __builtin_amdgcn_sched_barrier(0);                   // consume one, re-load one: without it hipcc gathers the 16 uses behind ONE vmcnt(0)
        }
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) acc ^= r[i];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[1] = 1u;       // keeps the loads alive
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = dep + 1u;
}
__global__ __launch_bounds__(512) void empty_kernel(const unsigned *__restrict__ in, unsigned *__restrict__ out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = in[0] + 1u;
}

int main(int argc, char **argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 224;          // 32 layers x 7 kernels
    // every launch reads a FRESH slice of a 6 GiB arena (the step's weights are 15 GB, read once per step: nothing a launch reads is in L2 or in the
    // 256 MiB memory-side cache from the launch before)
    const size_t arena = (size_t)6 << 30;
    v4u *W;
    unsigned *io;
    CK(hipMalloc(&W, arena));
    CK(hipMemset(W, 1, arena));
    CK(hipMalloc(&io, 64));
    CK(hipMemset(io, 0, 64));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double mbs[] = {0, 4.2, 16.8, 33.6, 50.3, 63, 117.4, 235.3};
    printf("# %d dependent launches per measurement, 256 workgroups x 512 threads (8 waves, 16 KiB per wave in flight), fresh bytes per launch\n", launches);
    double t0us = 0;
    for (double mb : mbs) {
        long iters = (long)(mb * 1e6 / (256.0 * 512 * 16) + 0.5);
        iters = (iters + PF - 1) / PF * PF;
        const size_t bytes = (size_t)iters * 256 * 512 * 16, slots = bytes ? arena / bytes : 1;
        for (int rep = 0; rep < 2; ++rep) {                      // first repetition warms up
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < launches; ++i) {
                if (iters) hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(512), 0, st, W + (size_t)(i % slots) * (bytes / 16), iters, io, io);
                else hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, st, io, io);
            }
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            if (rep == 1) {
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / launches;
                if (iters == 0) t0us = us;
                printf("%8.1f MB: %7.2f us per launch", bytes / 1e6, us);
                if (iters) printf("  = %5.2f TB/s over the launch;  beyond the empty launch (%.2f us): %6.2f us = %5.2f TB/s", bytes / us / 1e6, t0us, us - t0us, bytes / (us - t0us) / 1e6);
                printf("\n");
            }
        }
    }
    return 0;
}
