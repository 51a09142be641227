// Which physical XCDs / CUs does a CU-masked stream (hipExtStreamCreateWithCUMask) run on?  Census of HW_REG_XCC_ID / HW_ID over 4096 spinning workgroups
// for the two mask shapes tools/probe_cu_mask.py uses.   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/cu_mask_probe tools/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void census(unsigned *out, int spin) {
    unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf;      // HW_REG_XCC_ID[3:0]
    unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));           // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | (hwid & 0xffff);
}

static int run(const char *name, const std::vector<int> &bits) {
    unsigned mask[8] = {0};
    for (int b : bits) mask[b >> 5] |= 1u << (b & 31);
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, 8, mask));
    const int nb = 4096;
    unsigned *out;
    CK(hipMalloc(&out, nb * 4));
    hipLaunchKernelGGL(census, dim3(nb), dim3(64), 0, st, out, 2000);
    CK(hipStreamSynchronize(st));
    std::vector<unsigned> h(nb);
    CK(hipMemcpy(h.data(), out, nb * 4, hipMemcpyDeviceToHost));
    int per_xcc[16] = {0};
    std::set<unsigned> units;
    for (unsigned v : h) { per_xcc[v >> 16]++; units.insert(((v >> 16) << 16) | ((v >> 8) & 0xff)); }     // (xcc, se | sh | cu)
    printf("%-44s %3zu mask bits -> %3zu distinct (XCC, SE, SH, CU) units; workgroups per XCC:", name, bits.size(), units.size());
    for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
    printf("\n");
    CK(hipStreamDestroy(st));
    CK(hipFree(out));
    return 0;
}

int main() {
    std::vector<int> all, xcd3, low96, comp3;
    for (int i = 0; i < 256; ++i) {
        all.push_back(i);
        if (i % 8 < 3) xcd3.push_back(i); else comp3.push_back(i);
        if (i / 8 < 12) low96.push_back(i);
    }
    if (run("all 256 bits", all)) return 1;
    if (run("bits with i % 8 < 3 (XCDs 0-2 whole?)", xcd3)) return 1;
    if (run("bits with i % 8 >= 3 (the other five?)", comp3)) return 1;
    if (run("bits with i / 8 < 12 (12 CUs of every XCD?)", low96)) return 1;
    return 0;
}
