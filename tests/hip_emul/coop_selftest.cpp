// Self-test of the shim's cooperative launch (tests/hip_emul/hip_emul.h): blocks of one launch run concurrently over shared
// "device" memory, so a hand-rolled grid barrier completes and data published before it is visible after it.
#include <hip/hip_runtime.h>

__global__ void coop_kernel(unsigned *counter, unsigned *slots, unsigned *out, int rounds) {
    __shared__ unsigned seen;
    const unsigned nb = gridDim.x, b = blockIdx.x;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) __hip_atomic_store(slots + b, (unsigned)(r * 1000 + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {                                   // grid barrier: monotonic counter, target (r + 1) * nb
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r + 1) * nb) __builtin_amdgcn_s_sleep(1);
            seen = __hip_atomic_load(slots + (b + 1) % nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (threadIdx.x == 1) out[r * nb + b] = seen;              // the neighbour's value of THIS round
        __syncthreads();
        if (threadIdx.x == 0) {                                   // second barrier: nobody overwrites its slot before all have read
            __hip_atomic_fetch_add(counter + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r + 1) * nb) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}

extern "C" int coop_selftest(int blocks, int threads, int rounds, unsigned *out_host) {
    unsigned *counter = nullptr, *slots = nullptr, *out = nullptr;
    if (hipMalloc((void **)&counter, 64) || hipMalloc((void **)&slots, blocks * 4) || hipMalloc((void **)&out, (size_t)rounds * blocks * 4)) return -1;
    hipMemset(counter, 0, 64);
    hipMemset(slots, 0xff, blocks * 4);
    hipMemset(out, 0xff, (size_t)rounds * blocks * 4);
    void *params[] = {&counter, &slots, &out, &rounds};
    const hipError_t e = hipLaunchCooperativeKernel(coop_kernel, dim3(blocks), dim3(threads), params, 0, nullptr);
    hipMemcpy(out_host, out, (size_t)rounds * blocks * 4, hipMemcpyDeviceToHost);
    hipFree(counter); hipFree(slots); hipFree(out);
    return (int)e;
}
