// hip_emul.h — HIP-on-threads shim (TEST INFRASTRUCTURE ONLY; never built into, shipped with or imported by the product).
//
// Resolved instead of <hip/hip_runtime.h> when a source is compiled with `-x c++ -I tests/hip_emul`: the SOURCE of the
// engine's kernels and of its host code then becomes a plain host program, so that index arithmetic, epilogue wiring,
// flag / tag protocols, launch sequencing and rounding points can be exercised in the `-m "not gpu"` suite at toy sizes:
//   * a kernel launch runs its blocks ONE AT A TIME, every thread of a block is an OS thread; a COOPERATIVE launch
//     (hipLaunchCooperativeKernel) runs every block in its own forked process, concurrently, so grid barriers work;
//   * `__shared__` is a plain static (static shared memory) — `extern __shared__` arrays are defined by the harness;
//   * `__syncthreads()` is a barrier over the block, wave shuffles and the 16x16x32 MFMA exchange their operands through a
//     per-wave buffer between two wave barriers (so they must be called wave-uniformly, as on the hardware);
//   * device memory is host memory, streams and events do nothing, every call is synchronous.
// What it cannot show: anything about the GPU memory model (scopes, caches, cross-device visibility), the order in which
// the matrix core adds its 32 products (fp32, sequential here), timing, occupancy or register pressure.
#pragma once
#define VLO_HIP_EMUL 1
#define ENG_SPIN_TICKS 60000000000ll      // csrc/gemv_engine.inc: bounded spins of the loader / consumer waves — OS threads are slow, never time out here
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <fcntl.h>
#include <stdio.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

// ---- language ---------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define address_space(x)                 // __attribute__((address_space(1))) -> no attribute on the host

typedef enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotSupported = 801 } hipError_t;
typedef void *hipStream_t;

struct emul_dim3 {
    unsigned x = 1, y = 1, z = 1;
    emul_dim3() {}
    emul_dim3(unsigned a, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
typedef emul_dim3 dim3;
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(8) ushort4 { unsigned short x, y, z, w; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }

struct emul_wave_ctx {
    pthread_barrier_t bar;
    unsigned slot[64], slot2[64];        // shuffles, row swaps
    float A[16 * 32], B[32 * 16];        // MFMA operands
};
struct emul_block_ctx {
    pthread_barrier_t block_bar;
    std::vector<emul_wave_ctx> waves;
};
static thread_local emul_dim3 threadIdx, blockIdx, blockDim, gridDim;
static thread_local emul_block_ctx *emul_ctx = nullptr;

// ---- direct-to-LDS loads (global_load_lds) ---------------------------------------------------------------------------------
// Two landing models, chosen with VLO_EMUL_GLDS: "sync" (default) — the bytes land when the instruction issues, the EARLIEST the
// hardware allows (a restage that comes too early overwrites data another thread still reads: TSan sees the race); "late" — the
// bytes land only when the issuing thread executes the s_waitcnt vmcnt(N) that retires the load (oldest first, N stay in flight),
// the LATEST the hardware allows: a fragment read that is not ordered behind the covering vmcnt + barrier reads stale LDS and the
// parity tests fail deterministically.  The build patches `asm volatile("s_waitcnt vmcnt(N)")` into emul_vmcnt(N).
struct emul_glds_op { void *dst; const void *src; int size; };
static thread_local std::vector<emul_glds_op> emul_glds_q;
static inline bool emul_glds_late() {
    static const bool late = getenv("VLO_EMUL_GLDS") && !strcmp(getenv("VLO_EMUL_GLDS"), "late");
    return late;
}
static inline void emul_vmcnt(int n) {
    size_t done = 0;
    while (emul_glds_q.size() - done > (size_t)n) {
        const emul_glds_op &o = emul_glds_q[done++];
        memcpy(o.dst, o.src, (size_t)o.size);
    }
    if (done) emul_glds_q.erase(emul_glds_q.begin(), emul_glds_q.begin() + (long)done);
}
static inline void emul_glds(const void *g, void *l, int size) {
    if (emul_glds_late()) emul_glds_q.push_back({l, g, size});
    else memcpy(l, g, (size_t)size);
}
// __syncthreads() = s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier (hipcc drains pending LDS writes before the workgroup barrier);
// the raw __builtin_amdgcn_s_barrier() does not wait for anything
static inline void emul_raw_barrier() { pthread_barrier_wait(&emul_ctx->block_bar); }
static inline void __syncthreads() { emul_vmcnt(0); emul_raw_barrier(); }

template <class T>
static inline T emul_shfl_from(T v, int src_lane_of_me /* computed from my lane */) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    emul_wave_ctx &W = emul_ctx->waves[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    memcpy(&W.slot[lane], &v, 4);
    pthread_barrier_wait(&W.bar);
    T r;
    memcpy(&r, &W.slot[src_lane_of_me], 4);
    pthread_barrier_wait(&W.bar);
    return r;
}
template <class T> static inline T __shfl_xor(T v, int lane_mask, int /*width*/ = 64) { return emul_shfl_from(v, (int)((threadIdx.x & 63) ^ lane_mask)); }
template <class T> static inline T __shfl_up(T v, unsigned delta, int /*width*/ = 64) {
    const int lane = threadIdx.x & 63;
    return emul_shfl_from(v, lane >= (int)delta ? lane - (int)delta : lane);
}
template <class T> static inline T __shfl(T v, int src, int /*width*/ = 64) { return emul_shfl_from(v, src & 63); }
// v_permlane16_swap / v_permlane32_swap (gfx950): the ODD rows (of 16 / 32 lanes) of the first operand trade places with the EVEN
// rows of the second one; returns {first, second} after the swap
typedef unsigned emul_u32x2 __attribute__((ext_vector_type(2)));
static inline emul_u32x2 emul_permlane_swap(unsigned vdst, unsigned src, int row) {
    emul_wave_ctx &W = emul_ctx->waves[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63, odd = (lane / row) & 1;
    W.slot[lane] = vdst;
    W.slot2[lane] = src;
    pthread_barrier_wait(&W.bar);
    emul_u32x2 r = {odd ? W.slot2[lane - row] : vdst, odd ? src : W.slot[lane + row]};
    pthread_barrier_wait(&W.bar);
    return r;
}
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) emul_permlane_swap((a), (b), 16)
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) emul_permlane_swap((a), (b), 32)
// wave vote: every lane publishes its predicate, every lane ORs the 64 slots
static inline int __any(int pred) {
    emul_wave_ctx &W = emul_ctx->waves[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const int p = pred ? 1 : 0;
    memcpy(&W.slot[lane], &p, 4);
    pthread_barrier_wait(&W.bar);
    int r = 0;
    for (int i = 0; i < 64; ++i) { int q; memcpy(&q, &W.slot[i], 4); r |= q; }
    pthread_barrier_wait(&W.bar);
    return r;
}

// v_mfma_f32_16x16x32_{bf16,f16}: D[16x16] = A[16x32] . B[32x16] + C, register maps of csrc/common.cuh:
//   A[m][k]: lane l holds m = l & 15, k = (l >> 4) * 8 + j      B[k][n]: lane l holds n = l & 15, k = (l >> 4) * 8 + j
//   D[m][n]: lane l holds n = l & 15, m = (l >> 4) * 4 + r
template <class V8, class V4>
static inline V4 emul_mfma_16x16x32(V8 a, V8 b, V4 c) {
    emul_wave_ctx &W = emul_ctx->waves[threadIdx.x >> 6];
    const int l = threadIdx.x & 63, i = l & 15, kb = (l >> 4) * 8;
    for (int j = 0; j < 8; ++j) {
        W.A[i * 32 + kb + j] = (float)a[j];
        W.B[(kb + j) * 16 + i] = (float)b[j];
    }
    pthread_barrier_wait(&W.bar);
    V4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int m = (l >> 4) * 4 + r;
        float s = c[r];
        for (int k = 0; k < 32; ++k) s += W.A[m * 32 + k] * W.B[k * 16 + i];
        d[r] = s;
    }
    pthread_barrier_wait(&W.bar);
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emul_mfma_16x16x32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) emul_mfma_16x16x32((a), (b), (c))

// v_mfma_f32_16x16x128_f8f6f4 on e4m3 bytes (cbsz = blgp = 0, unscaled): a lane's 32 operand bytes are row / column l & 15 and k slots
// (l >> 4) * 32 + [0, 32) — four passes of the 16x16x32 scratch.  fp32 accumulation here (the hardware's adder keeps fewer bits: DESIGN.md).
static inline float emul_fp8_e4m3(unsigned b);
template <class V8I, class V4>
static inline V4 emul_mfma_fp8_16x16x128(V8I a, V8I b, V4 c) {
    emul_wave_ctx &W = emul_ctx->waves[threadIdx.x >> 6];
    const int l = threadIdx.x & 63, i = l & 15, kb = (l >> 4) * 8;
    V4 d = c;
    for (int p = 0; p < 4; ++p) {
        for (int j = 0; j < 8; ++j) {
            const int byte = p * 8 + j;
            W.A[i * 32 + kb + j] = emul_fp8_e4m3(((unsigned)a[byte >> 2] >> ((byte & 3) * 8)) & 0xffu);
            W.B[(kb + j) * 16 + i] = emul_fp8_e4m3(((unsigned)b[byte >> 2] >> ((byte & 3) * 8)) & 0xffu);
        }
        pthread_barrier_wait(&W.bar);
        for (int r = 0; r < 4; ++r) {
            const int m = (l >> 4) * 4 + r;
            float s = d[r];
            for (int k = 0; k < 32; ++k) s += W.A[m * 32 + k] * W.B[k * 16 + i];
            d[r] = s;
        }
        pthread_barrier_wait(&W.bar);
    }
    return d;
}
#define __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, cbsz, blgp, oa, sa, ob, sb) emul_mfma_fp8_16x16x128((a), (b), (c))

static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
#define __expf(x) expf(x)                // glibc declares __expf but does not export it
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __frcp_rn(x) (1.0f / (x))
static inline long long wall_clock64() {      // s_memrealtime: a 100 MHz counter
    return (long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10);
}

#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), (order))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))
static inline float atomicAdd(float *p, float v) {          // relaxed fp32 atomic add (CAS loop)
    unsigned *u = reinterpret_cast<unsigned *>(p), old = __atomic_load_n(u, __ATOMIC_RELAXED);
    for (;;) {
        const unsigned want = __float_as_uint(__uint_as_float(old) + v);
        if (__atomic_compare_exchange_n(u, &old, want, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return __uint_as_float(old);
    }
}
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
#define __builtin_amdgcn_s_sleep(x) sched_yield()
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_s_barrier() emul_raw_barrier()
// a wave is in lockstep on the hardware (the builtin emits no instruction); here its 64 lanes are OS threads: rendezvous
#define __builtin_amdgcn_wave_barrier() pthread_barrier_wait(&emul_ctx->waves[threadIdx.x >> 6].bar)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)      /* instruction-scheduling fence: nothing to order on the CPU */
// OCP e4m3fn pair -> two floats (v_cvt_pk_f32_fp8): bytes 0,1 (sel = false) or 2,3 of src
static inline float emul_fp8_e4m3(unsigned b) {
    const unsigned e = (b >> 3) & 15u, m = b & 7u;
    float v = e == 0 ? ldexpf((float)m, -9) : ((e == 15 && m == 7) ? NAN : ldexpf(1.0f + (float)m / 8.0f, (int)e - 7));
    return (b & 0x80u) ? -v : v;
}
typedef float emul_f32x2 __attribute__((ext_vector_type(2)));
static inline emul_f32x2 emul_cvt_pk_f32_fp8(int src, bool hi) {
    const unsigned u = (unsigned)src >> (hi ? 16 : 0);
    emul_f32x2 r = {emul_fp8_e4m3(u & 0xffu), emul_fp8_e4m3((u >> 8) & 0xffu)};
    return r;
}
#define __builtin_amdgcn_cvt_pk_f32_fp8(src, sel) emul_cvt_pk_f32_fp8((src), (sel))
// v_cvt_pk_fp8_f32: two floats -> two e4m3fn bytes (round to nearest even, saturating at 448) into the low / high half of `old`
static inline unsigned emul_f32_to_e4m3(float f) {
    const unsigned sign = std::signbit(f) ? 0x80u : 0u;
    float a = fabsf(f);
    if (std::isnan(a)) return sign | 0x7fu;
    if (a > 448.f) a = 448.f;
    int ex;
    (void)frexpf(a, &ex);                                   // a = m 2^ex, m in [0.5, 1)
    const int step_exp = (ex - 4) < -9 ? -9 : (ex - 4);     // 8 steps per binade, 2^-9 below 2^-6
    const float q = ldexpf(nearbyintf(ldexpf(a, -step_exp)), step_exp);     // default rounding mode: ties to even
    if (q == 0.f) return sign;
    int e2;
    const float m = frexpf(q, &e2);                          // q = m 2^e2
    if (e2 - 1 < -6) return sign | (unsigned)lrintf(ldexpf(q, 9));          // subnormal: q / 2^-9
    return sign | ((unsigned)(e2 - 1 + 7) << 3) | (unsigned)lrintf((m * 2.f - 1.f) * 8.f);
}
static inline int emul_cvt_pk_fp8_f32(float a, float b, int old, bool hi) {
    const unsigned pair = emul_f32_to_e4m3(a) | (emul_f32_to_e4m3(b) << 8);
    return hi ? (int)(((unsigned)old & 0x0000ffffu) | (pair << 16)) : (int)(((unsigned)old & 0xffff0000u) | pair);
}
#define __builtin_amdgcn_cvt_pk_fp8_f32(a, b, old, sel) emul_cvt_pk_fp8_f32((a), (b), (old), (sel))
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(order)
// HW_REG_XCC_ID: a scrambled, uneven block -> "XCD" map (5 populated groups), so nothing may rely on a placement pattern
#define __builtin_amdgcn_s_getreg(imm) ((unsigned)((blockIdx.x * 5u + 3u) % 7u % 5u))
#define __builtin_nontemporal_load(p) (*(p))

// direct-to-LDS load: every lane's `size` bytes land at the wave-uniform LDS base + lane * size
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) emul_glds((const void *)(g), (char *)(l) + (threadIdx.x & 63) * (size), (size))

// ---- stream capture / graphs: launches and copies issued while the host thread captures are recorded, a graph launch replays them
struct emul_graph { std::vector<std::function<void()>> ops; };
typedef emul_graph *hipGraph_t;
typedef emul_graph *hipGraphExec_t;
inline thread_local emul_graph *emul_capturing = nullptr;

// ---- kernel launches --------------------------------------------------------------------------------------------------
// `body` = the kernel call with its arguments bound.  blockDim.x OS threads are created per launch and walk the blocks of
// the grid together, one block at a time (static shared memory is a process-wide static).
static inline void emul_launch(emul_dim3 grid, emul_dim3 block, const std::function<void()> &body) {
    if (emul_capturing) {
        emul_capturing->ops.push_back([grid, block, body]() { emul_launch(grid, block, body); });
        return;
    }
    const unsigned nt = block.x * block.y * block.z, nw = (nt + 63) / 64;
    if (nt == 0 || grid.x * grid.y * grid.z == 0) return;
    emul_block_ctx ctx;
    pthread_barrier_t end_bar;
    pthread_barrier_init(&ctx.block_bar, nullptr, nt);
    pthread_barrier_init(&end_bar, nullptr, nt);
    ctx.waves = std::vector<emul_wave_ctx>(nw);
    for (unsigned w = 0; w < nw; ++w) pthread_barrier_init(&ctx.waves[w].bar, nullptr, (w + 1) * 64 <= nt ? 64 : nt - w * 64);
    std::vector<std::thread> ts;
    ts.reserve(nt);
    for (unsigned t = 0; t < nt; ++t)
        ts.emplace_back([&, t]() {
            threadIdx = emul_dim3(t);            // 1-D blocks only (all of this tree's kernels)
            blockDim = block;
            gridDim = grid;
            emul_ctx = &ctx;
            for (unsigned bz = 0; bz < grid.z; ++bz)
                for (unsigned by = 0; by < grid.y; ++by)
                    for (unsigned bx = 0; bx < grid.x; ++bx) {
                        blockIdx = emul_dim3(bx, by, bz);
                        body();
                        emul_vmcnt(0);                       // a wave ends only when its loads have landed
                        pthread_barrier_wait(&end_bar);      // the block is over: its shared memory may be reused
                    }
        });
    for (auto &th : ts) th.join();
    pthread_barrier_destroy(&ctx.block_bar);
    pthread_barrier_destroy(&end_bar);
    for (auto &w : ctx.waves) pthread_barrier_destroy(&w.bar);
}
// Cooperative launches: the blocks must make progress TOGETHER (grid barriers, flags between blocks), so every block runs in
// its own forked process (own static shared memory, own thread set) over the shared "device" memory.
inline hipError_t emul_coop_error = hipSuccess;
static inline void emul_launch_concurrent(emul_dim3 grid, emul_dim3 block, const std::function<void()> &body) {
    fflush(nullptr);
    std::vector<pid_t> kids;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                const pid_t pid = fork();
                if (pid == 0) {
                    // this process is block (bx, by, bz): one block, the usual thread-per-lane execution
                    const unsigned nt = block.x, nw = (nt + 63) / 64;
                    emul_block_ctx ctx;
                    pthread_barrier_init(&ctx.block_bar, nullptr, nt);
                    ctx.waves = std::vector<emul_wave_ctx>(nw);
                    for (unsigned w = 0; w < nw; ++w) pthread_barrier_init(&ctx.waves[w].bar, nullptr, (w + 1) * 64 <= nt ? 64 : nt - w * 64);
                    std::vector<std::thread> ts;
                    for (unsigned t = 0; t < nt; ++t)
                        ts.emplace_back([&, t]() {
                            threadIdx = emul_dim3(t);
                            blockIdx = emul_dim3(bx, by, bz);
                            blockDim = block;
                            gridDim = grid;
                            emul_ctx = &ctx;
                            body();
                            emul_vmcnt(0);
                        });
                    for (auto &th : ts) th.join();
                    _exit(0);
                }
                if (pid < 0) { emul_coop_error = hipErrorOutOfMemory; break; }
                kids.push_back(pid);
            }
    for (pid_t k : kids) {
        int st = 0;
        if (waitpid(k, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) emul_coop_error = hipErrorInvalidValue;
    }
}
template <class... A, size_t... I>
static inline void emul_call_unpacked(void (*f)(A...), void **params, std::index_sequence<I...>) {
    f(*reinterpret_cast<std::remove_reference_t<A> *>(params[I])...);
}
template <class... A>
static inline hipError_t hipLaunchCooperativeKernel(void (*f)(A...), emul_dim3 grid, emul_dim3 block, void **params, unsigned, void *) {
    emul_launch_concurrent(grid, block, [=]() { emul_call_unpacked(f, params, std::index_sequence_for<A...>{}); });
    const hipError_t e = emul_coop_error;
    emul_coop_error = hipSuccess;
    return e;
}

#define hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, ...) \
    emul_launch(emul_dim3(grid), emul_dim3(block), [=]() { kernel(__VA_ARGS__); })

// ---- runtime API (device memory = host memory, one "device", everything synchronous) -------------------------------------
typedef struct emul_event { std::chrono::steady_clock::time_point t; } *hipEvent_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
typedef struct { char reserved[64]; } hipIpcMemHandle_t;
#define hipHostMallocDefault 0x0
#define hipHostMallocPortable 0x1
#define hipHostMallocMapped 0x2
#define hipDeviceMallocFinegrained 0x1
#define hipDeviceMallocUncached 0x3
#define hipIpcMemLazyEnablePeerAccess 0x1
typedef enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 } hipFuncAttribute;

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "emulated HIP runtime error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
typedef enum { hipDeviceAttributeMultiprocessorCount = 63 } hipDeviceAttribute_t;
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }     // as hipGetDeviceProperties below
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 4; return hipSuccess; }
static inline hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 0; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
// Every "device" allocation is a MAP_SHARED mapping: the block-processes of a cooperative launch (emul_launch_concurrent) and
// their parent see one memory.  Uncached / fine-grained allocations (the p2p mailboxes) are NAMED shared memory objects so
// that hipIpc can be emulated ACROSS independent processes: the "handle" is the object's name.
struct emul_shm_entry { std::string name; size_t bytes; bool owner; };
inline std::map<void *, emul_shm_entry> emul_shm_registry;
inline std::mutex emul_shm_mu;
static inline hipError_t hipMalloc(void **p, size_t bytes) {
    const size_t n = (bytes + 4095) / 4096 * 4096 + 4096;
    void *m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> g(emul_shm_mu);
    emul_shm_registry[m] = {"", n, true};
    *p = m;
    return hipSuccess;
}
static inline hipError_t hipExtMallocWithFlags(void **p, size_t bytes, unsigned) {
    static std::atomic<int> counter{0};
    char name[64];
    snprintf(name, sizeof(name), "/vlo_emul_%d_%d", (int)getpid(), counter++);
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return hipErrorOutOfMemory;
    if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name); return hipErrorOutOfMemory; }
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { shm_unlink(name); return hipErrorOutOfMemory; }
    std::lock_guard<std::mutex> g(emul_shm_mu);
    emul_shm_registry[m] = {name, bytes, true};
    *p = m;
    return hipSuccess;
}
static inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
static inline hipError_t hipFree(void *p) {
    {
        std::lock_guard<std::mutex> g(emul_shm_mu);
        auto it = emul_shm_registry.find(p);
        if (it != emul_shm_registry.end()) {
            munmap(p, it->second.bytes);
            if (it->second.owner && !it->second.name.empty()) shm_unlink(it->second.name.c_str());
            emul_shm_registry.erase(it);
            return hipSuccess;
        }
    }
    return p ? hipErrorInvalidValue : hipSuccess;
}
static inline hipError_t hipHostFree(void *p) { return hipFree(p); }
static inline hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) {
    if (emul_capturing) emul_capturing->ops.push_back([=]() { memset(p, v, n); });
    else memset(p, v, n);
    return hipSuccess;
}
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
    if (emul_capturing) emul_capturing->ops.push_back([=]() { memmove(d, s, n); });
    else memmove(d, s, n);
    return hipSuccess;
}
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind,
                                          hipStream_t) {
    auto op = [=]() { for (size_t r = 0; r < height; ++r) memmove((char *)d + r * dpitch, (const char *)s + r * spitch, width); };
    if (emul_capturing) emul_capturing->ops.push_back(op);
    else op();
    return hipSuccess;
}
static inline hipError_t hipMemcpy2D(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
    for (size_t r = 0; r < height; ++r) memmove((char *)d + r * dpitch, (const char *)s + r * spitch, width);
    return hipSuccess;
}
typedef enum { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 } hipStreamCaptureMode;
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) {
    if (emul_capturing) return hipErrorInvalidValue;
    emul_capturing = new emul_graph();
    return hipSuccess;
}
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g) {
    *g = emul_capturing;
    emul_capturing = nullptr;
    return *g ? hipSuccess : hipErrorInvalidValue;
}
static inline hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t g, void *, void *, size_t) { *e = new emul_graph(*g); return hipSuccess; }
static inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) {
    for (auto &op : e->ops) op();
    return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
#define hipStreamNonBlocking 1
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // emulated streams are synchronous
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emul_event(); return hipSuccess; }
#define hipEventDisableTiming 0x2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new emul_event(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *h, void *p) {
    std::lock_guard<std::mutex> g(emul_shm_mu);
    auto it = emul_shm_registry.find(p);
    if (it == emul_shm_registry.end() || it->second.name.empty() || it->second.name.size() >= sizeof(h->reserved)) return hipErrorInvalidValue;
    memset(h->reserved, 0, sizeof(h->reserved));
    memcpy(h->reserved, it->second.name.c_str(), it->second.name.size());
    return hipSuccess;
}
static inline hipError_t hipIpcOpenMemHandle(void **p, hipIpcMemHandle_t h, unsigned) {
    char name[65];
    memcpy(name, h.reserved, 64);
    name[64] = 0;
    const int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return hipErrorInvalidValue;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return hipErrorInvalidValue; }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> g(emul_shm_mu);
    emul_shm_registry[m] = {name, (size_t)st.st_size, false};
    *p = m;
    return hipSuccess;
}
static inline hipError_t hipIpcCloseMemHandle(void *p) { return hipFree(p); }
