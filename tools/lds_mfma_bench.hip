// lds_mfma_bench.hip -- dev tool: how do ds_read_b128 / ds_write_b128 streams and MFMA streams share a CU?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lds_mfma_bench.hip -o tools/lds_mfma_bench && tools/lds_mfma_bench
// One 512-thread block per CU (waves w and w + 4 share a SIMD), ITERS iterations; per iteration a "compute" wave
// issues 16 v_mfma_f32_32x32x16_f16 (512 matrix-pipe cycles) and a "memory" wave 12 conflict-free ds_read_b128
// (+ optionally 4 ds_write_b128): the per-half-K-tile mix of k_gemm256.
//   mode 0  waves 0..3 MFMA, waves 4..7 exit                      (matrix pipe alone: 512 cycles / iteration)
//   mode 1  all 8 waves MFMA                                        (1024 / iteration)
//   mode 2  waves 0..3 MFMA, waves 4..7 reads                      (do the partner's reads slow the MFMAs?)
//   mode 3  waves 0..3 MFMA, waves 4..7 reads + writes
//   mode 4  waves 4..7 reads only, 0..3 exit                        (LDS alone)
//   mode 5  all 8 waves: reads of iteration i + 1 issued before the 16 MFMAs of iteration i, which consume them
//           (software pipeline, no barrier)
//   mode 6  as 5 plus 4 ds_write_b128 per iteration
//   mode 7  as 5 with an s_barrier per iteration
//   mode 8  two groups one phase apart (ping-pong): [reads, drain, barrier, 16 MFMAs, barrier]
//   mode 9  as 8 plus the 4 writes in the memory phase
//   mode 10 as 9 plus 4 buffer_load_dwordx4 per memory phase (the stage writes store what was requested one
//           iteration earlier: hand-counted vmcnt) -- the full memory phase of k_gemm256
//   mode 11 as 10, but waves 2, 3 (6, 7) run [writes, loads, reads] while waves 0, 1 (4, 5) run [reads, writes, loads]
//   mode 12 as 10 with the order [writes, loads, reads] for every wave
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Frag { half8 a[4]; half8 b[2]; };

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float *sink, unsigned long long *ticks, int iters, const unsigned char *src) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool grpY = wave >= 4;
    for (int i = tid; i < 32768; i += 512) ((unsigned int *)smem)[i] = 0x3c003c00u;   // halves 1.0
    __syncthreads();
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // fragment addresses as in k_gemm256 (128-byte rows, 16-byte chunk c at c ^ ((row >> 1) & 7))
    const int wm = (wave >> 2) & 1, wn = wave & 3;
    auto rd = [&](int ks, Frag &f) {
        const _Float16 *sA = (const _Float16 *)smem, *sB = (const _Float16 *)(smem + 32768);
        const int c = ks * 2 + (lane >> 5);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wn * 64 + j * 32 + (lane & 31);
            f.b[j] = *(const half8 *)(sB + row * 64 + ((c ^ ((row >> 1) & 7)) << 3));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wm * 128 + i * 32 + (lane & 31);
            f.a[i] = *(const half8 *)(sA + row * 64 + ((c ^ ((row >> 1) & 7)) << 3));
        }
    };
    auto mma = [&](const Frag &f) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.b[j], f.a[i], acc[i][j], 0, 0, 0);
    };
    u32x4 wv = {1u, 2u, 3u, (unsigned)tid};
    u32x4 rg[4] = {wv, wv, wv, wv};
    auto wr = [&]() {
        unsigned char *st = smem + 65536;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = wave * 32 + q * 8 + (lane >> 3), c = lane & 7;
            *(u32x4 *)(st + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = MODE >= 10 ? rg[q] : wv;
        }
    };
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 1 << 20, 0x00020000);
    unsigned off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) off[q] = (unsigned)((wave * 32 + q * 8 + (lane >> 3)) * 1024 + (lane & 7) * 16);
    auto ld = [&](int it) {
        const int so = (it & 7) * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rg[q]) : "v"(off[q]), "s"(rs), "s"(so) : "memory");
    };
    Frag f0 = {}, f1 = {}, g0 = {}, g1 = {};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE <= 4) {
        const bool compute = MODE == 1 || (MODE != 4 && !grpY);
        const bool memory = (MODE == 2 || MODE == 3 || MODE == 4) && grpY;
        if (!compute && !memory) return;
        if (compute) {
            rd(0, f0);
            rd(1, f1);
            for (int it = 0; it < iters; ++it) {
                mma(f0);
                mma(f1);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
                rd(0, f0);
                rd(1, f1);
                if (MODE == 3) wr();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 4; ++i) { asm volatile("" ::"v"(f0.a[i])); asm volatile("" ::"v"(f1.a[i])); }
#pragma unroll
                for (int j = 0; j < 2; ++j) { asm volatile("" ::"v"(f0.b[j])); asm volatile("" ::"v"(f1.b[j])); }
            }
        }
    } else if (MODE <= 7) {
        rd(0, f0);
        rd(1, f1);
        for (int it = 0; it < iters; it += 2) {
            __builtin_amdgcn_sched_barrier(0);
            rd(2, g0);
            rd(3, g1);
            if (MODE == 6) wr();
            __builtin_amdgcn_sched_barrier(0);
            mma(f0);
            mma(f1);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
            __builtin_amdgcn_sched_barrier(0);
            rd(0, f0);
            rd(1, f1);
            if (MODE == 6) wr();
            __builtin_amdgcn_sched_barrier(0);
            mma(g0);
            mma(g1);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        }
    } else {
        if (MODE >= 10) ld(0);
        const bool rot = MODE == 12 || (MODE == 11 && (wave & 2));
        if (grpY) __builtin_amdgcn_s_barrier();
        for (int it = 0; it < iters; ++it) {
            if (MODE >= 10 && rot) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wr();
                ld(it + 1);
                __builtin_amdgcn_sched_barrier(0);
                rd((it & 1) * 2, f0);
                rd((it & 1) * 2 + 1, f1);
            } else {
                rd((it & 1) * 2, f0);
                rd((it & 1) * 2 + 1, f1);
                __builtin_amdgcn_sched_barrier(0);
                if (MODE >= 10) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (MODE >= 9) wr();
                if (MODE >= 10) ld(it + 1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            mma(f0);
            mma(f1);
            __builtin_amdgcn_sched_barrier(0);
            if (!(grpY && it == iters - 1)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s += (float)f0.a[0][0] + (float)f1.b[0][0] + (float)g0.a[1][0] + (float)g1.b[1][0] + (float)rg[0][0] + (float)rg[3][1];
    if (s == 12345.678f) sink[tid] = s;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
static void run(const char *what, float *sink, unsigned long long *dt, int iters, const unsigned char *src) {
    const int lds = 65536 + 65536;
    CK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(dt, 0, 256 * 8 * 8));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), lds, 0, sink, dt, iters, src);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), lds, 0, sink, dt, iters, src);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256 * 8];
    CK(hipMemcpy(h, dt, sizeof(h), hipMemcpyDeviceToHost));
    double sx = 0, sy = 0;
    for (int b = 0; b < 256; ++b) { sx += (double)h[b * 8]; sy += (double)h[b * 8 + 4]; }
    printf("mode %d  %-58s %8.1f us  per iteration: %6.1f ns, ticks wave0 %6.0f wave4 %6.0f\n", MODE, what, ms * 1e3, ms * 1e6 / iters,
           sx / 256 / iters, sy / 256 / iters);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    float *sink;
    unsigned long long *dt;
    CK(hipMalloc(&sink, 4096));
    CK(hipMalloc(&dt, 256 * 8 * 8));
    unsigned char *src;
    CK(hipMalloc(&src, 1 << 20));
    CK(hipMemset(src, 0, 1 << 20));
    run<0>("4 MFMA waves (16 MFMA / iteration)", sink, dt, iters, src);
    run<1>("8 MFMA waves", sink, dt, iters, src);
    run<2>("4 MFMA waves + 4 reader waves (12 ds_read_b128)", sink, dt, iters, src);
    run<3>("4 MFMA waves + 4 reader/writer waves (12 rd + 4 wr)", sink, dt, iters, src);
    run<4>("4 reader waves alone", sink, dt, iters, src);
    run<5>("8 waves, reads pipelined under own MFMAs", sink, dt, iters, src);
    run<6>("8 waves, reads + writes pipelined under own MFMAs", sink, dt, iters, src);
    run<7>("as 5 with drain + s_barrier per iteration", sink, dt, iters, src);
    run<8>("ping-pong groups: [12 rd, drain, barrier, 16 MFMA, barrier]", sink, dt, iters, src);
    run<9>("ping-pong groups with 4 writes in the memory phase", sink, dt, iters, src);
    run<10>("ping-pong, memory phase = 12 rd + 4 wr + 4 buffer loads", sink, dt, iters, src);
    run<11>("as 10, waves 2,3 / 6,7 in the order wr, ld, rd", sink, dt, iters, src);
    run<12>("as 10, every wave in the order wr, ld, rd", sink, dt, iters, src);
    return 0;
}
