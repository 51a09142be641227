// What does the SIZE of a kernel's argument block cost per launch?  Round 6 saw the gate/up GEMV lose 0.5 us when GemvArgs grew from 208 to 296 bytes
// (fields the kernel never reads).  Two shapes per size: an (almost) empty kernel, and a kernel that streams 64 MB (the step's kernels are neither
// empty nor large); dependent launches back to back on one stream, HIP events around 2000 of them.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/kernarg_size_probe tools/kernarg_size_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int BYTES> struct Args { float *p; const float4 *src; size_t n4; int use; char pad[BYTES - 28]; };

template <int BYTES>
__global__ __launch_bounds__(512) void probe_kernel(Args<BYTES> a) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = a.src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (a.use || acc == 123.456f) a.p[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)a.pad[BYTES - 29];      // the LAST byte of the block is read
}

template <int BYTES>
static int run(float *out, const float4 *src, hipStream_t st) {
    for (int big = 0; big < 2; ++big) {
        Args<BYTES> a{};
        a.p = out; a.src = src; a.n4 = big ? (size_t)(64 << 20) / 16 : 0; a.use = 0;
        const int iters = big ? 400 : 2000;
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe_kernel<BYTES>, dim3(256), dim3(512), 0, st, a);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(probe_kernel<BYTES>, dim3(256), dim3(512), 0, st, a);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("kernel-argument block %4d bytes, %s: %7.3f us per dependent launch\n", BYTES, big ? "64 MB streamed " : "no memory work ", ms * 1e3 / iters);
    }
    return 0;
}

int main() {
    float *out;
    float4 *src;
    CK(hipMalloc(&out, 256 * 512 * 4));
    CK(hipMalloc(&src, (size_t)(64 << 20)));
    CK(hipMemset(src, 0, (size_t)(64 << 20)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 2; ++rep) {
        if (run<64>(out, src, st) || run<128>(out, src, st) || run<208>(out, src, st) || run<256>(out, src, st) || run<296>(out, src, st) || run<512>(out, src, st) ||
            run<1024>(out, src, st))
            return 1;
        printf("--\n");
    }
    return 0;
}
