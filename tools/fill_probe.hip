// fill_probe.hip — what ONE CU can pull out of L2 into LDS (or registers), with every CU doing it at once: the bound of the one-frame ViT GEMMs
// (64 x 64 tiles move 113 - 151 MB through the vector memory path per launch for 7.5 MB of unique bytes).
// 256 workgroups (one per CU, or 2 x 256 threads), each wave loops over 1-KiB pieces (8 rows x 128 B of a row-major [rows][K] fp16 matrix — the GEMM tile
// access pattern — or fully contiguous KiBs) of a buffer small enough to stay in the XCD's L2 (every workgroup of an XCD re-reads the same `footprint`
// bytes, from different starting rows), by global_load_lds_dwordx4 under a counted vmcnt or by global_load_dwordx4 into registers.
// Output: bytes per clock per CU (at the measured wall time and a nominal 2.4 GHz) and chip-wide TB/s.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fill_probe.hip -o tools/_bin/fill_probe ; tools/_bin/fill_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// MODE 0: direct-to-LDS, DEPTH pieces in flight per wave; MODE 1: registers, DEPTH loads in flight per lane
// rows: the buffer is [rows][rowbytes]; piece p of a wave = rows 8 p' .. 8 p' + 7, 128 B at column offset c (strided form), or 1 KiB contiguous (rowbytes = 128)
template <int MODE, int DEPTH, int NT>
__global__ __launch_bounds__(NT) void fill_kernel(const char *__restrict__ buf, int rows, int rowbytes, long pieces_per_wave, unsigned *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) char lds[(NT / 64) * DEPTH * 1024];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = NT / 64;
    const int lrow = lane >> 3, lc = lane & 7;
    const int rowgroups = rows / 8, ccs = __builtin_ctz(rowbytes / 128);        // rowbytes / 128 is a power of two
    const int P = rowgroups << ccs;                                              // pieces of the buffer
    // a wave walks the pieces in steps of nw (the block's waves interleave), column chunk fastest: for a fixed 8-row group the K chunks in order, then the
    // next row group — the order a GEMM tile's K loop produces; all 32-bit and division-free (wave-uniform: scalar ALU)
    int k = (int)(((long)blockIdx.x * 7919 + (long)w) % P);
    v4u acc = {0u, 0u, 0u, 0u};
    v4u r[DEPTH];
    auto next = [&]() {
        const int rg = k >> ccs, cc = k & ((1 << ccs) - 1);
        const char *p = buf + ((size_t)(rg * 8 + lrow) * rowbytes + (size_t)cc * 128 + lc * 16);
        k += nw;
        if (k >= P) k -= P;
        return p;
    };
    if (MODE == 0) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)next(), (__attribute__((address_space(3))) void *)&lds[(w * DEPTH + d) * 1024], 16, 0, 0);
        for (long i = DEPTH; i < pieces_per_wave; i += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)next(), (__attribute__((address_space(3))) void *)&lds[(w * DEPTH + d) * 1024], 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc[0] = *reinterpret_cast<const unsigned *>(&lds[(w * DEPTH) * 1024 + lane * 16]);
    } else {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) r[d] = *reinterpret_cast<const v4u *>(next());
        for (long i = DEPTH; i < pieces_per_wave; i += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                acc ^= r[d];
                r[d] = *reinterpret_cast<const v4u *>(next());
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= r[d];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[0] = 1u;
}

template <int MODE, int DEPTH, int NT>
static void run(const char *name, const char *buf, size_t footprint, int rowbytes, int blocks, unsigned *out, hipStream_t st) {
    const int rows = (int)(footprint / rowbytes);
    const long ppw = 4096 / (NT / 256 ? NT / 256 : 1);            // pieces per wave: 4 MiB per wave ... per block NT/64 waves
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((fill_kernel<MODE, DEPTH, NT>), dim3(blocks), dim3(NT), 0, st, buf, rows, rowbytes, ppw, out);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        if (rep == 1) {
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (double)blocks * (NT / 64) * ppw * 1024.0;
            const double tbs = bytes / (ms * 1e-3) / 1e12;
            printf("%-34s footprint %7.2f MiB rowbytes %5d blocks %3d x %3d thr: %7.3f ms  %6.2f TB/s chip  %5.1f B/clk/CU (256 CUs @ 2.4 GHz)\n", name, footprint / 1048576.0, rowbytes,
                   blocks, NT, ms, tbs, tbs * 1e12 / 256 / 2.4e9);
        }
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main() {
    char *buf;
    unsigned *out;
    const size_t cap = (size_t)512 << 20;
    CK(hipMalloc(&buf, cap));
    CK(hipMemset(buf, 1, cap));
    CK(hipMalloc(&out, 64));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const size_t fps[] = {(size_t)1 << 20, (size_t)2 << 20, (size_t)3 << 20, (size_t)16 << 20, (size_t)128 << 20, (size_t)512 << 20};
    for (size_t fp : fps) {
        for (int rowbytes : {2048, 128}) {
            run<0, 4, 512>("glds depth 4, 512 thr", buf, fp, rowbytes, 256, out, st);
            run<0, 8, 512>("glds depth 8, 512 thr", buf, fp, rowbytes, 256, out, st);
            run<0, 16, 512>("glds depth 16, 512 thr", buf, fp, rowbytes, 256, out, st);
            run<0, 8, 256>("glds depth 8, 2 x 256 thr per CU", buf, fp, rowbytes, 512, out, st);
            run<1, 8, 512>("regs depth 8, 512 thr", buf, fp, rowbytes, 256, out, st);
            run<1, 16, 512>("regs depth 16, 512 thr", buf, fp, rowbytes, 256, out, st);
            run<1, 8, 1024>("regs depth 8, 1024 thr", buf, fp, rowbytes, 256, out, st);
        }
    }
    return 0;
}
