// mfma_class_probe.hip -- what does the register class of an MFMA accumulator cost on gfx950?  (dev tool, round 4)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_class_probe.hip -o tools/mfma_class_probe
// One wave per SIMD (256 threads, one block per CU), 16 independent 32x32x16 f16 accumulators, NV of them pinned to VGPRs and
// the rest to AGPRs by asm constraints; and the dependent chain (one accumulator, back to back) in either class.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NV>
__global__ __launch_bounds__(256) void k_indep(float *out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.01f); }
    f32x16 acc[16];
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t >= 16 - NV) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
    float s = 0.f;
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VG, int NACC>   // chain: NACC accumulators used round-robin (dependency distance NACC MFMAs), class VG ? VGPR : AGPR
__global__ __launch_bounds__(256) void k_chain(float *out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.01f); }
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (VG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[k % NACC]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[k % NACC]) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *out;
    CK(hipMalloc(&out, 256 * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    auto run = [&](const char *name, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-58s %8.1f us  = %6.1f ns per MFMA\n", name, ms * 1e3, ms * 1e6 / (iters * 16.0));
        return 0;
    };
    for (int blocks : {64, 256}) {
        printf("-- %d blocks of 4 waves (one per SIMD), %d x 16 MFMAs per wave\n", blocks, iters);
        run("16 independent accumulators, all AGPR", [&] { hipLaunchKernelGGL(k_indep<0>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        run("16 independent accumulators, 2 of them VGPR", [&] { hipLaunchKernelGGL(k_indep<2>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        run("16 independent accumulators, 8 of them VGPR", [&] { hipLaunchKernelGGL(k_indep<8>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        run("16 independent accumulators, all VGPR", [&] { hipLaunchKernelGGL(k_indep<16>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        run("dependent chain, 1 AGPR accumulator", [&] { hipLaunchKernelGGL((k_chain<0, 1>), dim3(blocks), dim3(256), 0, 0, out, iters); });
        run("dependent chain, 1 VGPR accumulator", [&] { hipLaunchKernelGGL((k_chain<1, 1>), dim3(blocks), dim3(256), 0, 0, out, iters); });
        run("2 VGPR accumulators alternating", [&] { hipLaunchKernelGGL((k_chain<1, 2>), dim3(blocks), dim3(256), 0, 0, out, iters); });
        run("4 VGPR accumulators round-robin", [&] { hipLaunchKernelGGL((k_chain<1, 4>), dim3(blocks), dim3(256), 0, 0, out, iters); });
        run("2 AGPR accumulators alternating", [&] { hipLaunchKernelGGL((k_chain<0, 2>), dim3(blocks), dim3(256), 0, 0, out, iters); });
    }
    return 0;
}
