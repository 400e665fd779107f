// barrier_bench.hip -- dev micro-benchmark: what does one s_barrier cost a K-loop?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/barrier_bench.hip -o tools/barrier_bench
// Blocks of W waves loop { s_barrier; K x v_mfma_f32_32x32x16_f16 (waves < MW only) } N times; the
// per-iteration time minus K x 32 cycles of matrix-pipe time is the barrier's cost in that setting.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K>
__global__ __launch_bounds__(1024) void k_bar(int n, int mw, float *sink, unsigned long long *tm) {
    extern __shared__ char sm[];
    const int wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    half8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {2, 1, 2, 1, 2, 1, 2, 1};
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        __builtin_amdgcn_s_barrier();
        if (wave < mw) {
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k & 3], 0, 0, 0);
        }
    }
    float t = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (t == 123.456f) sink[0] = t + sm[0];
    if (threadIdx.x == 0 && tm) tm[blockIdx.x] = wall_clock64() - t0;
}

template <int K>
static void run(int waves, int mw, int blocks_per_cu, float *sink, unsigned long long *tm) {
    const int n = 256, grid = 256 * blocks_per_cu;
    const size_t lds = blocks_per_cu == 1 ? 96 << 10 : 64 << 10;
    CK(hipFuncSetAttribute((const void *)k_bar<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_bar<K>, dim3(grid), dim3(waves * 64), lds, 0, n, mw, sink, tm);
    CK(hipDeviceSynchronize());
    unsigned long long h[512];
    CK(hipMemcpy(h, tm, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
    double us = 0;
    for (int b = 0; b < grid; ++b) us += h[b] / 100.0;
    us /= grid;
    printf("waves/block %2d (MFMA waves %d) blocks/CU %d  MFMAs per barrier %2d: %7.1f ns per iteration (MFMA time alone %6.1f ns @2.1 GHz)\n",
           waves, mw, blocks_per_cu, K, us * 1e3 / n, K * 32 / 2.1);
}

int main() {
    float *sink; unsigned long long *tm;
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&tm, 8 * 512));
    for (int bpc = 1; bpc <= 2; ++bpc)
        for (int waves : {4, 8, 16}) {
            if (waves == 16 && bpc == 2) continue;
            const int mw = waves >= 8 ? waves / 2 : waves;
            run<0>(waves, mw, bpc, sink, tm); run<4>(waves, mw, bpc, sink, tm); run<16>(waves, mw, bpc, sink, tm); run<32>(waves, mw, bpc, sink, tm);
        }
    return 0;
}
