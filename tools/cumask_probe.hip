// cumask_probe.hip -- dev tool: which XCC / SE / CU does bit i of a hipExtStreamCreateWithCUMask mask select?
//   hipcc --offload-arch=gfx950 -O3 tools/cumask_probe.hip -o tools/cumask_probe && tools/cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k(unsigned *out, int spin) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = hwid; }
}

static void run(const char *what, const std::vector<uint32_t> &mask) {
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    const int nb = 4096;
    unsigned *d;
    CK(hipMalloc(&d, nb * 8));
    hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, s, d, 2000);   // 20 us per block: the grid spreads over every allowed CU
    CK(hipStreamSynchronize(s));
    std::vector<unsigned> h(nb * 2);
    CK(hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int b = 0; b < nb; ++b) {
        const unsigned xcc = h[b * 2] & 0xF, hw = h[b * 2 + 1];
        const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[xcc].insert(se * 32 + sh * 16 + cu);
    }
    printf("%-44s:", what);
    int total = 0;
    for (auto &kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  = %d CUs\n", total);
    CK(hipFree(d));
    CK(hipStreamDestroy(s));
}

int main() {
    std::vector<uint32_t> all(8, 0xFFFFFFFFu);
    run("all 256 bits", all);
    for (int w = 0; w < 8; ++w) {
        std::vector<uint32_t> m(8, 0);
        m[w] = 0xFFFFFFFFu;
        char buf[64];
        snprintf(buf, sizeof buf, "bits %d..%d", w * 32, w * 32 + 31);
        run(buf, m);
    }
    {
        std::vector<uint32_t> m(8, 0);
        for (int i = 0; i < 256; i += 8) m[i / 32] |= 1u << (i % 32);
        run("every 8th bit (0, 8, 16, ...)", m);
    }
    {
        std::vector<uint32_t> m(8, 0);
        for (int i = 0; i < 256; ++i) if ((i % 8) < 2) m[i / 32] |= 1u << (i % 32);
        run("bits with i % 8 in {0, 1}", m);
    }
    {
        std::vector<uint32_t> m(8, 0);
        for (int i = 0; i < 64; ++i) m[i / 32] |= 1u << (i % 32);
        run("bits 0..63", m);
    }
    {
        std::vector<uint32_t> m(8, 0);
        for (int i = 0; i < 128; ++i) m[i / 32] |= 1u << (i % 32);
        run("bits 0..127", m);
    }
    return 0;
}
