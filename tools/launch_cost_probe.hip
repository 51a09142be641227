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

// MODE 0: bare stream.  MODE 1: every 2 x PF loads per thread (32 KiB per wave = one "group" of the GEMV: two column tiles) the 8 waves leave a float4
// partial in LDS, meet at ONE __syncthreads and two waves reduce the 8 partials — the GEMV's group boundary (gemv_epi.inc) without its arithmetic.
// MODE 2: the same partials, but no block barrier: a wave bumps an LDS counter after writing its partial, whoever arrives LAST reduces (fixed wave
// order: deterministic), the others stream on (ring of 4 partial buffers).
template <int MODE>
__global__ __launch_bounds__(512) void stream_kernel(const v4u *__restrict__ W, long iters, const unsigned *__restrict__ in, unsigned *__restrict__ out) {
    __shared__ float4 part[4][8][64];
    __shared__ unsigned arrived[4];
    const unsigned dep = in[0];                                  // the previous launch's word: a true dependency
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const v4u *wp = W + (size_t)blockIdx.x * iters * 512 + threadIdx.x;
    v4u acc = {dep, 0u, 0u, 0u};
    v4u r[PF];
    if (MODE == 2 && threadIdx.x < 4) arrived[threadIdx.x] = 0;
    if (MODE == 2) __syncthreads();
    float keep = 0.f;
    // iters is a multiple of 2 PF (host): no load of the loop sits behind a predicate
#pragma unroll
    for (int i = 0; i < PF; ++i) r[i] = __builtin_nontemporal_load(wp + (size_t)i * 512);
    int g = 0;
    for (long i0 = PF; i0 < iters; i0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            acc ^= r[i];
            r[i] = __builtin_nontemporal_load(wp + (size_t)(i0 + i) * 512);
        }
        if (MODE != 0 && ((i0 / PF) & 1) == 0) {                // a group boundary: the next group's first loads are already in flight
            const int b = g & 3;
            part[b][w][lane] = make_float4(__uint_as_float(acc[0]), __uint_as_float(acc[1]), __uint_as_float(acc[2]), __uint_as_float(acc[3]));
            if (MODE == 1) {
                __syncthreads();
                if (w < 2) {
                    float4 t = part[b][0][lane];
#pragma unroll
                    for (int ww = 1; ww < 8; ++ww) { const float4 u = part[b][ww][lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
                    keep += t.x + t.y + t.z + t.w;
                }
            } else {
                unsigned n = 0;
                if (lane == 0) n = __hip_atomic_fetch_add(&arrived[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                n = __builtin_amdgcn_readfirstlane(n);
                if (n == 7u) {                                    // last of the 8 waves: everybody's partial is in LDS (a wave's LDS operations retire in order)
                    float4 t = part[b][0][lane];
#pragma unroll
                    for (int ww = 1; ww < 8; ++ww) { const float4 u = part[b][ww][lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
                    keep += t.x + t.y + t.z + t.w;
                    if (lane == 0) arrived[b] = 0;
                }
            }
            ++g;
        }
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) acc ^= r[i];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u || keep == 1.2345f) out[1] = 1u;       // keeps the loads and the reductions alive
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
    const double mbs[] = {0, 33.6, 50.3, 67.1, 117.4, 235.3};
    printf("# %d dependent launches per measurement, 256 workgroups x 512 threads (8 waves, 16 KiB per wave in flight), fresh bytes per launch\n", launches);
    const char *names[3] = {"bare stream", "+ barrier and reduce per 32 KiB per wave (the GEMV's group boundary)", "+ last-arriving wave reduces, no barrier"};
    for (int mode = 0; mode < 3; ++mode) {
        printf("-- %s\n", names[mode]);
        double t0us = 0;
        for (double mb : mbs) {
            long iters = (long)(mb * 1e6 / (256.0 * 512 * 16) + 0.5);
            iters = (iters + 2 * PF - 1) / (2 * PF) * (2 * PF);
            const size_t bytes = (size_t)iters * 256 * 512 * 16, slots = bytes ? arena / bytes : 1;
            for (int rep = 0; rep < 2; ++rep) {                  // first repetition warms up
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < launches; ++i) {
                    const v4u *wsrc = W + (size_t)(i % slots) * (bytes / 16);
                    if (!iters) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, st, io, io);
                    else if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(256), dim3(512), 0, st, wsrc, iters, io, io);
                    else if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(256), dim3(512), 0, st, wsrc, iters, io, io);
                    else hipLaunchKernelGGL(stream_kernel<2>, dim3(256), dim3(512), 0, st, wsrc, iters, io, io);
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
    }
    return 0;
}
