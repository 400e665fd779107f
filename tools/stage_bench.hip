// stage_bench.hip -- dev micro-benchmark: what does the L2 -> LDS direct-load path of one CU sustain?
//
// Every block runs LW loader waves.  A loader wave streams 1 KB pieces (global_load_lds_dwordx4, 16 B
// per lane) of an L2-resident region into its own LDS ring, keeping DEPTH pieces in flight (counted
// vmcnt).  Two source patterns: PAT 0 = the GEMM's A-tile piece (8 rows x 128 B, row pitch 1 KB, chunks
// XOR-swizzled), PAT 1 = 1 KB contiguous.  CONS consumer waves run ds_read_b128 + MFMA next to the
// loaders (no synchronisation) to show what the K loop's other half costs the loaders.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stage_bench.hip -o tools/stage_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// region: 256 KB per block slot, 32 slots (8 MB): XCD x (blocks x, x+8, ...) touches 4 slots = 1 MB of its 4 MB L2
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: global_load_lds_dwordx4; MODE 1: buffer_load_dwordx4 ... lds (SGPR descriptor, 32-bit offsets);
// MODE 2: register staging (global_load_dwordx4 -> 4 VGPRs -> ds_write_b128), 8 pieces per batch, the
// next batch's loads issued before the current batch is written (DEPTH is then the LDS ring only)
template <int LW, int DEPTH, int PAT, int CONS, int MODE = 0, int CK = 0, int PRIO = 0>
__global__ __launch_bounds__((LW + CONS) * 64) void k_stage(const unsigned char *src, int pieces, float *sink, unsigned long long *tm, int cons_steps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned char *region = src + (size_t)(blockIdx.x & 31) * (256 << 10);
    const unsigned long long t0 = wall_clock64();
    struct Stamp { unsigned long long *tm, t0; int slot, on; __device__ ~Stamp() { if (on && tm) tm[slot] = wall_clock64() - t0; } };
    // per block: [0] loader wave 0 finish, [1] consumer wave 0 finish (100 MHz ticks since block start)
    Stamp stamp{tm, t0, (int)blockIdx.x * 2 + (wave < LW ? 0 : 1), (lane == 0 && (wave == 0 || wave == LW)) ? 1 : 0};
    if (wave < LW && PRIO) __builtin_amdgcn_s_setprio(PRIO);
    if (wave < LW && (MODE == 2 || MODE == 3)) {
        unsigned char *ring = smem + wave * DEPTH * 1024;
        auto addr = [&](int p) -> const u32x4 * {
            const int pp = p * LW + wave;
            if (PAT == 0) {
                const int chunk = pp & 31, kt = (pp >> 5) & 7;
                const int row = chunk * 8 + (lane >> 3);
                return (const u32x4 *)(region + (size_t)row * 1024 + kt * 128 + (lane & 7) * 16);
            }
            return (const u32x4 *)(region + (size_t)(pp & 255) * 1024 + lane * 16);
        };
        // inline asm on both sides: the compiler neither tracks these loads (explicit counted vmcnt) nor
        // removes the LDS writes; two register batches, no copies
        u32x4 ra[8], rb[8];
        const unsigned lbase = (unsigned)(uintptr_t)(ring) + lane * 16;
        __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc((void *)region, 0, 256 << 10, 0x00020000);
        auto ld = [&](u32x4 &r, int p) {
            if (MODE == 2) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(addr(p)) : "memory"); return; }
            // MODE 3: buffer_load_dwordx4 (SGPR descriptor, 32-bit VGPR offset) into registers
            const unsigned off = (unsigned)((const unsigned char *)addr(p) - region);
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r) : "v"(off), "s"(rs3) : "memory");
        };
        auto st = [&](const u32x4 &r, int p) {
            asm volatile("ds_write_b128 %0, %1" ::"v"(lbase + (unsigned)((p % DEPTH) * 1024)), "v"(r) : "memory");
        };
#pragma unroll
        for (int q = 0; q < 8; ++q) ld(ra[q], q);
        for (int p = 0; p < pieces; p += 16) {
#pragma unroll
            for (int q = 0; q < 8; ++q) ld(rb[q], p + 8 + q);
            wait_vm<8>();
#pragma unroll
            for (int q = 0; q < 8; ++q) st(ra[q], p + q);
#pragma unroll
            for (int q = 0; q < 8; ++q) ld(ra[q], p + 16 + q);
            wait_vm<8>();
#pragma unroll
            for (int q = 0; q < 8; ++q) st(rb[q], p + 8 + q);
        }
        wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (sink && lane == 0 && pieces < 0) sink[blockIdx.x] = ring[0];
    } else if (wave < LW && MODE == 1) {
        unsigned char *ring = smem + wave * DEPTH * 1024;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)region, 0, 256 << 10, 0x00020000);
        for (int p = 0; p < pieces; ++p) {
            const int pp = p * LW + wave;
            unsigned voff, soff;
            if (PAT == 0) {
                const int chunk = pp & 31, kt = (pp >> 5) & 7;
                const int row = (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);   // (chunk * 8 + row) >> 1 & 7 == row >> 1 & 7 ^ ... kept simple
                voff = (unsigned)(row * 1024 + c * 16);
                soff = (unsigned)(chunk * 8192 + kt * 128);
            } else {
                voff = (unsigned)(lane * 16);
                soff = (unsigned)((pp & 255) * 1024);
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(ring + (p % DEPTH) * 1024), 16, voff, soff, 0, 0);
            if (p >= DEPTH - 1) wait_vm<DEPTH - 1>();
        }
        wait_vm<0>();
        if (sink && lane == 0 && pieces < 0) sink[blockIdx.x] = ring[0];
    } else if (wave < LW) {
        unsigned char *ring = smem + wave * DEPTH * 1024;
        // piece p of this wave: PAT 0 -> rows (8 p' .. 8 p' + 7) of a [256][512] half matrix at k-step (p / 32) % 8
        for (int p = 0; p < pieces; ++p) {
            const int pp = p * LW + wave;
            const unsigned char *g;
            if (PAT == 0) {
                const int chunk = pp & 31, kt = (pp >> 5) & 7;
                const int row = chunk * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                g = region + (size_t)row * 1024 + kt * 128 + c * 16;
            } else {
                g = region + (size_t)(pp & 255) * 1024 + lane * 16;
            }
            glds16(g, ring + (p % DEPTH) * 1024);
            if (p >= DEPTH - 1) wait_vm<DEPTH - 1>();
        }
        wait_vm<0>();
        if (sink && lane == 0 && pieces < 0) sink[blockIdx.x] = ring[0];
    } else {
        // consumer: 16 MFMA + 8 ds_read_b128 per "K-step", as many K-steps as the loaders have pieces / 8
        const unsigned char *rd = smem + (tid & 255) * 16;
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const int steps = cons_steps > 0 ? cons_steps : pieces * LW / 32;   // default: one K-step of MFMAs per 32 KB staged (128 x 128 x 64 tile)
        // CK: 0 = ds_read_b128 + MFMA, 1 = MFMA only, 2 = ds_read only (+ cheap VALU), 3 = plain VALU FMAs only
        half8 c0 = *(const half8 *)rd, c1 = *(const half8 *)(rd + 4096);
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                half8 a0 = c0, a1 = c1;
                if (CK == 0 || CK == 2) { a0 = *(const half8 *)(rd + (ks * 2 + 0) * 4096); a1 = *(const half8 *)(rd + (ks * 2 + 1) * 4096); }
                if (CK == 4) {
                    // half duty: every MFMA is followed by ~32 idle cycles on this wave
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, a1, acc[0], 0, 0, 0);
                    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, a0, acc[1], 0, 0, 0);
                    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, a0, acc[2], 0, 0, 0);
                    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, a1, acc[3], 0, 0, 0);
                    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
                } else if (CK <= 1) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, a1, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, a0, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, a0, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, a1, acc[3], 0, 0, 0);
                } else if (CK == 2) {
                    acc[0][0] += (float)a0[0] + (float)a1[3];
                    asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc[0][r] = acc[0][r] * 1.0001f + 0.5f; acc[1][r] = acc[1][r] * 1.0001f + 0.5f;
                        acc[2][r] = acc[2][r] * 1.0001f + 0.5f; acc[3][r] = acc[3][r] * 1.0001f + 0.5f;
                    }
                }
            }
        }
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][7];
        if (sink && t == 12345.678f) sink[blockIdx.x] = t;
    }
}

static unsigned char *g_src;
static float *g_sink;
static unsigned long long *g_tm;
static int g_cons_steps = 0;
static hipEvent_t e0, e1;

template <int LW, int DEPTH, int PAT, int CONS, int MODE = 0, int CK = 0, int PRIO = 0>
static void run(int bpc, int pieces) {
    // LDS request decides how many blocks share a CU: 1 -> 96 KB, 2 -> 64 KB
    size_t lds = bpc == 1 ? 96 << 10 : 64 << 10;
    if ((size_t)LW * DEPTH * 1024 > lds) return;
    auto kern = k_stage<LW, DEPTH, PAT, CONS, MODE, CK, PRIO>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = 256 * bpc;
    CK(hipMemset(g_tm, 0, 1024 * 16));
    hipLaunchKernelGGL(kern, dim3(grid), dim3((LW + CONS) * 64), lds, 0, g_src, pieces, g_sink, g_tm, g_cons_steps);
    CK(hipDeviceSynchronize());
    unsigned long long htm[1024];
    CK(hipMemcpy(htm, g_tm, sizeof(htm), hipMemcpyDeviceToHost));
    double tl = 0, tc = 0;
    for (int b = 0; b < grid; ++b) { tl += htm[2 * b]; tc += htm[2 * b + 1]; }
    tl = tl / grid / 100.0; tc = tc / grid / 100.0;   // us at 100 MHz
    const int iters = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3((LW + CONS) * 64), lds, 0, g_src, pieces, g_sink, (unsigned long long *)nullptr, g_cons_steps);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    const double bytes = (double)grid * LW * pieces * 1024.0;
    printf("ck=%d prio=%d mode=%d LW=%d depth=%2d pat=%d cons=%d blocks/CU=%d  %8.2f us  %6.2f TB/s  %6.1f GB/s/CU  (%.1f B/clk/CU @2.4GHz)  in-block: loaders %.2f us, consumers %.2f us\n", CK, PRIO, MODE, LW, DEPTH, PAT,
           CONS, bpc, us, bytes / us / 1e6, bytes / us / 1e3 / 256, bytes / us / 1e3 / 256 / 2.4, tl, tc);
    fflush(stdout);
}

template <int LW, int PAT, int CONS>
static void sweep_depth(int bpc, int total_kb) {
    const int pieces = total_kb / LW;   // per wave: every block moves total_kb KB
    run<LW, 8, PAT, CONS>(bpc, pieces);
    run<LW, 8, PAT, CONS, 1>(bpc, pieces);
    run<LW, 8, PAT, CONS, 2>(bpc, pieces);
}

template <int MODE, int CK, int PRIO>
static void arb(int bpc) {
    run<4, 8, 0, 4, MODE, CK, PRIO>(bpc, 512);
}

int main(int argc, char **argv) {
    CK(hipMalloc(&g_src, 32 * (256 << 10)));
    CK(hipMemset(g_src, 1, 32 * (256 << 10)));
    CK(hipMalloc(&g_sink, 4096));
    CK(hipMalloc(&g_tm, 1024 * 16));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int kb = 2048;   // KB moved per block
    if (argc > 1) {
        // arbitration study: 4 loader waves + 4 consumer waves, one block per CU; which consumer activity
        // stalls the loaders, and does a raised loader priority help?
        printf("---- loaders alone\n");
        run<4, 8, 0, 0, 0>(1, 512); run<4, 8, 0, 0, 1>(1, 512); run<4, 8, 0, 0, 2>(1, 512);
        g_cons_steps = 256;   // consumers outlast the loaders: the loaders' in-block time is their rate under load
        printf("---- consumers alone (256 K-steps; loaders move 16 KB only)\n");
        run<4, 8, 0, 4, 0, 0>(1, 4); run<4, 8, 0, 4, 0, 1>(1, 4); run<4, 8, 0, 4, 0, 2>(1, 4); run<4, 8, 0, 4, 0, 3>(1, 4); run<4, 8, 0, 4, 0, 4>(1, 4);
        printf("---- both: 4 loader waves move 2 MB per block under 256 K-steps of consumers\n");
        arb<0, 0, 0>(1); arb<0, 1, 0>(1); arb<0, 2, 0>(1); arb<0, 3, 0>(1); arb<0, 4, 0>(1);
        arb<1, 0, 0>(1); arb<1, 1, 0>(1); arb<1, 4, 0>(1);
        arb<2, 0, 0>(1); arb<2, 1, 0>(1); arb<2, 2, 0>(1); arb<2, 3, 0>(1); arb<2, 4, 0>(1);
        printf("---- mode 3: buffer_load_dwordx4 -> VGPR -> ds_write_b128\n");
        g_cons_steps = 0;
        run<4, 8, 0, 0, 3>(1, 512); run<2, 8, 0, 0, 3>(1, 1024); run<8, 8, 0, 0, 3>(1, 256);
        g_cons_steps = 256;
        arb<3, 0, 0>(1); arb<3, 1, 0>(1); arb<3, 2, 0>(1); arb<3, 4, 0>(1);
        run<2, 8, 0, 4, 3, 1, 0>(1, 1024); run<8, 8, 0, 4, 3, 1, 0>(1, 256); run<8, 8, 0, 4, 3, 0, 0>(1, 256);
        run<8, 8, 0, 4, 1, 1, 0>(1, 256); run<8, 8, 0, 4, 1, 0, 0>(1, 256);
        printf("---- both, loaders at s_setprio 3\n");
        arb<0, 1, 3>(1); arb<2, 1, 3>(1);
        return 0;
    }
    for (int bpc = 1; bpc <= 2; ++bpc) {
        printf("---- loaders only, A-tile pattern\n");
        sweep_depth<1, 0, 0>(bpc, kb); sweep_depth<2, 0, 0>(bpc, kb); sweep_depth<4, 0, 0>(bpc, kb); sweep_depth<8, 0, 0>(bpc, kb);
        printf("---- loaders only, contiguous 1 KB pieces\n");
        sweep_depth<1, 1, 0>(bpc, kb); sweep_depth<4, 1, 0>(bpc, kb); sweep_depth<8, 1, 0>(bpc, kb);
        printf("---- loaders + 4 consumer waves (ds_read_b128 + MFMA), A-tile pattern\n");
        sweep_depth<1, 0, 4>(bpc, kb); sweep_depth<2, 0, 4>(bpc, kb); sweep_depth<4, 0, 4>(bpc, kb); sweep_depth<8, 0, 4>(bpc, kb);
        printf("---- loaders + 4 consumer waves, contiguous pieces\n");
        sweep_depth<2, 1, 4>(bpc, kb); sweep_depth<4, 1, 4>(bpc, kb);
    }
    return 0;
}
