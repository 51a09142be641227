// prefetch_probe.hip — does pulling the NEXT kernel's first weight bytes into L2 during the drain of the current kernel shorten a
// chain of dependent weight-streaming kernels?  Same synthetic decoder layer as chain_probe.hip (7 dependent kernels streaming the
// layer's byte counts on one stream).  Variant: in its last round of loads every block also loads the first `pfk` x 8 KiB of the
// bytes the same block index streams first in the next kernel (plain loads: the lines stay in the XCD's L2; block i of both
// kernels runs on XCD i % 8).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/prefetch_probe tools/prefetch_probe.hip ; run: tools/prefetch_probe [layers] [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define PF 16              // uint4 in flight per thread (16 KiB per wave)

template <int PFK>
__global__ __launch_bounds__(512) void stage(const v4u *__restrict__ W, int iters, const float *__restrict__ in, float *__restrict__ out, int n_io,
                                             const v4u *__restrict__ Wnext, int iters_next, int blocks_next) {
    const int tid = threadIdx.x;
    const v4u *wp = W + (size_t)blockIdx.x * iters * 512 + tid;
    v4u r[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i)
        if (i < iters) r[i] = __builtin_nontemporal_load(wp + (size_t)i * 512);
    float acc = 0.f;
    const float x = in[tid % n_io];                 // the producer's output
    v4u pf[PFK > 0 ? PFK : 1];
    for (int i0 = 0; i0 < iters; i0 += PF) {
        if (PFK > 0 && i0 + PF >= iters && Wnext && (int)blockIdx.x < blocks_next) {       // last round: nothing of this kernel left to issue
            const v4u *np = Wnext + (size_t)blockIdx.x * iters_next * 512 + tid;
#pragma unroll
            for (int i = 0; i < PFK; ++i)
                if (i < iters_next) pf[i] = np[(size_t)i * 512];
        }
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            if (i0 + i < iters) {
                const v4u v = r[i];
                acc += __uint_as_float((v.x ^ v.y ^ v.z ^ v.w) & 0x007fffffu) * 1e-30f * x;
                if (i0 + i + PF < iters) r[i] = __builtin_nontemporal_load(wp + (size_t)(i0 + i + PF) * 512);
            }
        }
    }
    if (PFK > 0 && Wnext && (int)blockIdx.x < blocks_next) {
#pragma unroll
        for (int i = 0; i < PFK; ++i)
            if (i < iters_next) acc += __uint_as_float(pf[i].x & 0x007fffffu) * 1e-38f;
    }
    for (int j = blockIdx.x * 512 + tid; j < n_io; j += gridDim.x * 512) out[j] = in[j] + 1.0f + (acc != acc ? 1.f : 0.f);
}

struct Stage { const char *name; double mb; int blocks; };

template <int PFK>
static double run(int layers, int reps, v4u *W, size_t pool_bytes, float **buf, int n_io, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int graph = 0, int empty = 0, int gemv_only = 0) {
    const Stage st[7] = {{"add_rmsnorm", 0.5, 16}, {"qkv", 50.3, 256}, {"attention", 27.0, 256}, {"combine", 2.0, 64},
                         {"o_proj", 33.5, 256}, {"gate_up", 235.0, 256}, {"down", 117.0, 256}};
    double best = 1e30;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipMemset(buf[0], 0, n_io * 4));
        CK(hipDeviceSynchronize());
        size_t woff = 0;
        int seq = 0;
        auto geom = [&](int k, int *iters, size_t *need) {
            int it = (int)(st[k].mb * 1e6 / 16 / 512 / st[k].blocks);
            if (it < 1) it = 1;
            *iters = it;
            *need = (size_t)st[k].blocks * it * 512;
        };
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        if (graph) CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        else CK(hipEventRecord(e0, s));
        for (int l = 0; l < layers; ++l)
            for (int k = 0; k < 7; ++k, ++seq) {
                int iters, iters_n;
                size_t need, need_n;
                geom(k, &iters, &need);
                geom((k + 1) % 7, &iters_n, &need_n);
                if (empty) iters = 0;
                if (woff + need > pool_bytes / 16) woff = 0;
                size_t woff_n = woff + need;
                if (woff_n + need_n > pool_bytes / 16) woff_n = 0;
                // gemv_only: the transitions the engine could wire: o_proj -> gate_up, gate_up -> down, down -> (add_rmsnorm) -> qkv
                const v4u *pfw = W + woff_n;
                int pf_it = iters_n, pf_bl = st[(k + 1) % 7].blocks;
                if (gemv_only) {
                    if (k == 4 || k == 5) { /* next stage */ }
                    else if (k == 6) {                      // skip the small kernel in between: its bytes come first in the pool
                        int it1; size_t need1;
                        geom(1, &it1, &need1);
                        size_t w1 = woff_n + need_n;
                        if (w1 + need1 > pool_bytes / 16) w1 = 0;
                        pfw = W + w1; pf_it = it1; pf_bl = st[1].blocks;
                    } else pfw = nullptr;
                }
                hipLaunchKernelGGL(stage<PFK>, dim3(st[k].blocks), dim3(512), 0, s, W + woff, iters, buf[seq & 1], buf[(seq + 1) & 1], n_io,
                                   pfw, pf_it, pf_bl);
                woff = woff_n;
            }
        if (graph) {
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s));                 // warm
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
        }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        if (graph) { CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    float v;
    CK(hipMemcpy(&v, buf[(layers * 7) & 1], 4, hipMemcpyDeviceToHost));
    printf("%s%s%sprefetch %2d x 8 KiB per block (%5.1f MB per kernel): %d layers x 7 kernels: %.3f ms = %.1f us per layer; chain value %.0f (expect %d)\n",
           graph ? "[hipGraph] " : "", empty ? "[EMPTY kernels] " : "", gemv_only ? "[GEMV->GEMV transitions only] " : "", PFK, PFK * 8.0 * 256 / 1024, layers, best, best * 1e3 / layers, v, layers * (graph ? 14 : 7));
    return best;
}

int main(int argc, char **argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 32, reps = argc > 2 ? atoi(argv[2]) : 5;
    const size_t pool_bytes = (size_t)3 << 30;
    v4u *W;
    CK(hipMalloc(&W, pool_bytes));
    CK(hipMemset(W, 0x11, pool_bytes));
    const int n_io = 11 * 4096;
    float *buf[2];
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&buf[i], n_io * 4)); CK(hipMemset(buf[i], 0, n_io * 4)); }
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    run<0>(layers, reps, W, pool_bytes, buf, n_io, s, e0, e1);
    run<2>(layers, reps, W, pool_bytes, buf, n_io, s, e0, e1);
    run<4>(layers, reps, W, pool_bytes, buf, n_io, s, e0, e1);
    run<8>(layers, reps, W, pool_bytes, buf, n_io, s, e0, e1);
    run<16>(layers, reps, W, pool_bytes, buf, n_io, s, e0, e1);
    run<0>(layers, reps, W, pool_bytes, buf, n_io, s, e0, e1);
    run<1>(layers, reps, W, pool_bytes, buf, n_io, s, e0, e1, 0, 0, 1);
    run<2>(layers, reps, W, pool_bytes, buf, n_io, s, e0, e1, 0, 0, 1);
    run<4>(layers, reps, W, pool_bytes, buf, n_io, s, e0, e1, 0, 0, 1);
    run<8>(layers, reps, W, pool_bytes, buf, n_io, s, e0, e1, 0, 0, 1);
    run<0>(layers, reps, W, pool_bytes, buf, n_io, s, e0, e1, 1);
    run<2>(layers, reps, W, pool_bytes, buf, n_io, s, e0, e1, 1);
    run<0>(layers, reps, W, pool_bytes, buf, n_io, s, e0, e1, 0, 1);
    run<0>(layers, reps, W, pool_bytes, buf, n_io, s, e0, e1, 1, 1);
    return 0;
}
