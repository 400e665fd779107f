// qv_gemm.hip -- fused-epilogue f16 GEMM on v_mfma_f32_32x32x16_f16 (see qv_kernels.h).
//
// C[M,N] = epilogue(A[M,K] * W[N,K]^T + bias).  128 x BN x 64 tiles (BN = 128 or 64), 512 threads: 4 consumer waves
// (2 x 2, fragment reads + MFMA + epilogue) and 4 loader waves (register-staged MUBUF loads into a 2-stage LDS
// ring by default; direct global->LDS loads with 2..4 stages under QVERSE_GEMM_LD=0).  The large shapes run on the
// 256 x 256-tile kernel of qv_gemm256.hip instead (launch_gemm's plan, bottom of this file); both kernels produce
// bit-identical results.  The epilogue goes back through LDS so that every
// global store is a full 16-byte lane-contiguous row segment (a wave writes 4 whole tile rows per
// instruction); storing straight from the MFMA accumulator layout (one row per lane) cost more
// time than the whole K loop.

#include "qv_kernels.h"
#include "qv_gemm_dequant.h"
#include "qv_ort.h"

#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>
#include <vector>

namespace {

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

__device__ __forceinline__ void glds16(const void *g, void *l) {
    // direct global->LDS, 16 B per lane; LDS destination = wave-uniform base + lane * 16
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

#ifdef QV_GEMM_TRACE
#define QV_ABL(bit) (g.abl & (bit))
#define QV_PHASE(slot) do { if (g.phase && tid == 0) g.phase[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (slot)] = wall_clock64(); } while (0)
#define QV_TRACE(slot) do { if (g.trace && lane == 0 && (wave & 3) == 0) g.trace[(((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 64 + kt_) * 4 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define QV_ABL(bit) false
#define QV_PHASE(slot) do { } while (0)
#define QV_TRACE(slot) do { } while (0)
#endif

constexpr bool epi_is_f32(int epi) { return epi == EPI_RESID || epi == EPI_F32; }

}  // namespace

// Wave specialisation: 512 threads.  Waves 0..3 (one per SIMD) are CONSUMERS: fragment reads from LDS
// and MFMAs only (2 x 2 layout, 64 x BN/2 each), then the epilogue.  Waves 4..7 are LOADERS: they only
// issue the direct global->LDS loads of the stage NST - 1 K-steps ahead.  A wave that does both pays
// the issue time of its 8 loads (hundreds of cycles per K-step) in front of its 16 MFMAs; split, the
// loads issue under the partner wave's MFMAs (tools/gemm_exp.hip: 7-17 % per GEMM at these shapes).
//
// LD selects how the loader waves move a K-step's operands into LDS:
//   LD = 0  direct global->LDS loads (global_load_lds_dwordx4), NST LDS stages, counted vmcnt
//   LD = 1  REGISTER staging through MUBUF: buffer_load_dwordx4 into VGPRs (a loader wave holds no
//           accumulators, so it has ~100 VGPRs to spare) and ds_write_b128 into a 2-stage LDS ring; the
//           prefetch depth (two K-steps) lives in registers, not in LDS.  Why (tools/stage_bench.hip, 4 loader
//           + 4 MFMA waves per CU): while a SIMD's matrix pipe runs back-to-back MFMAs, the FLAT-encoded
//           loads of its other waves (global_load_lds_dwordx4 AND global_load_dwordx4 -- their 64-bit
//           address goes through the VALU) hardly issue at all: 2 MB per CU took 86 / 74 us under 61 us of
//           dense MFMAs against 27 / 17 us alone, i.e. the loads ran AFTER the MFMAs.  Buffer loads (SGPR
//           descriptor + 32-bit offset, no VALU pass) are not blocked: 18.8 us under the same MFMA load
//           (53 B/clk/CU), and the MFMA waves lose < 2 %.
template <int EPI, int BN, int WQ, int NST, int LD>
__global__ __launch_bounds__(512, 4) void k_gemm(GemmArgs g) {
    constexpr bool W4 = WQ == 4, W8 = WQ == 8;
    // A8W8 (QV_PREC_ORT_MIXED): both operands are bytes and K counts byte PAIRS, so loaders, LDS image and fragment
    // reads are the f16 kernel's unchanged -- a 16-byte fragment chunk is 16 k values for v_mfma_i32_32x32x32_i8
    // instead of 8 halves -- and only the accumulator type and the epilogue differ
    constexpr bool I8 = WQ == 88;
    static_assert(!W8 || BN == 128, "W8A16 is built for 128-wide tiles only");
    static_assert(!I8 || (BN == 128 && LD == 1), "A8W8 is built for 128-wide tiles with register-staged loaders only");
    constexpr int BM = 128, BK = 64;
    constexpr int NT = 512;
    constexpr int WN = BN / 2;   // columns per consumer wave
    constexpr int NF = WN / 32;  // 32-wide B fragments per consumer wave
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = W4 ? BN * BK / 2 : W8 ? BN * BK : BN * BK * 2;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int G = W4 ? 5 : W8 ? 6 : 4 + BN / 32;      // direct loads per loader wave per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    QV_PHASE(0);
    const bool loader = wave >= 4;
    const int w4 = wave & 3;                     // index inside the consumer / loader group
    const int wm = w4 >> 1, wn = w4 & 1;
    // XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8); give each XCD
    // a contiguous run of tiles that share the A row panel so its private L2 sees the reuse.
    const int gx = gridDim.x, nwg = gridDim.x * gridDim.y;
    int wg = blockIdx.y * gx + blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (wg / gx) * BM, n0 = (wg % gx) * BN;

    typedef int i32x16 __attribute__((ext_vector_type(16)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef typename std::conditional<I8, i32x16, f32x16>::type acc_t;
    acc_t acc[2][NF];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    // LDS tile rows are 128 B (64 halves); the 16-B chunk index is XORed with (row >> 1) & 7 --
    // on the global SOURCE address (the direct load writes LDS lane-linearly) and on the fragment
    // read -- which spreads each ds_read_b128 lane group over all 16 slots of the 256-B bank row.
    const int nk = g.K / BK;
    auto stage = [&](int kt) {
        half_t *sA = (half_t *)(smem + (kt % NST) * STAGE_BYTES), *sB = (half_t *)((unsigned char *)sA + A_BYTES);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int chunk = w4 * 4 + q;
            int row = chunk * 8 + (lane >> 3);
            int c = (lane & 7) ^ ((row >> 1) & 7);
            int grow = m0 + row;
            grow = grow < g.M ? grow : g.M - 1;
            glds16(g.A + (size_t)grow * g.lda + kt * BK + c * 8, sA + chunk * 512);
        }
        if (W4) {
            // 64 x 64 nibble tiles are 2 KB contiguous in HBM and already swizzled: one load per 32
            // tile rows.  Every loader wave issues exactly one (with 64-wide tiles waves 2, 3 repeat
            // the loads of waves 0, 1) so that the vmcnt bookkeeping is the same for all of them.
            const int p = w4 & (BN / 32 - 1);
            glds16(g.Wq + ((size_t)((n0 >> 6) + (p >> 1)) * nk + kt) * 2048 + (p & 1) * 1024 + lane * 16,
                   (unsigned char *)sB + p * 1024);
            return;
        }
        if (W8) {
            // 64 x 64 byte tiles, 4 KB contiguous and already swizzled: two 1 KB pieces per loader wave
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int p = w4 * 2 + q;
                glds16(g.W8 + ((size_t)((n0 >> 6) + (p >> 2)) * nk + kt) * 4096 + (p & 3) * 1024 + lane * 16,
                       (unsigned char *)sB + p * 1024);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < BN / 32; ++q) {
            int chunk = w4 * (BN / 32) + q;
            int row = chunk * 8 + (lane >> 3);
            int c = (lane & 7) ^ ((row >> 1) & 7);
            glds16(g.W + (size_t)(n0 + row) * g.ldw + kt * BK + c * 8, sB + chunk * 512);
        }
    };

    // W4: this tile's scales [K/128][BN] live in LDS behind the operand stages for the whole K loop
    half2_t *sS = (half2_t *)(smem + NST * STAGE_BYTES);   // {scale, 1024 + zero point}
    if (W4) {
        const int nkb = g.K >> 7;
        for (int idx = tid; idx < BN * nkb; idx += NT) {
            int kb = idx / BN, n = idx - kb * BN;
            sS[idx] = ((const half2_t *)g.wscale)[(size_t)kb * g.N + n0 + n];
        }
        __syncthreads();
    }

    // One raw s_barrier per K-step joins the two groups.  A loader arrives once ITS loads of stage
    // kt have landed (counted vmcnt: the loads of the younger stages stay in flight), which makes
    // the stage visible to the consumers and tells the loaders that the consumers are done with
    // stage kt - 1, whose buffer the next prefetch overwrites.  (__syncthreads() would drain the
    // whole queue and serialise the load latency with the MFMAs.)
    if (loader && LD == 1) {
        static_assert(LD == 0 || NST == 2, "register staging uses a 2-stage LDS ring");
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        constexpr int NB = W4 ? 1 : W8 ? 2 : BN / 32;   // B pieces (1 KB each) per loader wave per K-step
        constexpr int GR = 4 + NB;
        // MUBUF addressing: SGPR descriptor + per-lane 32-bit byte offset (fixed for the tile) + a scalar
        // byte offset that advances with the K-step -- no VALU instruction per load (see the header comment)
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)g.A, 0, (int)((size_t)g.M * g.lda * 2), 0x00020000);
        const void *bbase = W4 ? (const void *)g.Wq : W8 ? (const void *)g.W8 : I8 ? (const void *)g.Wi8 : (const void *)g.W;
        const size_t bbytes = W4 ? (size_t)g.N * g.K / 2 : W8 ? (size_t)g.N * g.K : (size_t)g.N * g.ldw * 2;
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)bbase, 0, (int)bbytes, 0x00020000);
        unsigned offA[4], offB[NB];
        int dstA[4], dstB[NB];
        int stepB;   // bytes between consecutive K-steps of a B piece
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int chunk = w4 * 4 + q, row = chunk * 8 + (lane >> 3), c = lane & 7;
            int grow = m0 + row;
            grow = grow < g.M ? grow : g.M - 1;
            offA[q] = (unsigned)(((size_t)grow * g.lda + c * 8) * 2);
            dstA[q] = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
        }
        if (W4) {
            const int p = w4 & (BN / 32 - 1);
            offB[0] = (unsigned)(((size_t)((n0 >> 6) + (p >> 1)) * nk) * 2048 + (p & 1) * 1024 + lane * 16);
            dstB[0] = A_BYTES + p * 1024 + lane * 16;
            stepB = 2048;
        } else if (W8) {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int p = w4 * 2 + q;
                offB[q] = (unsigned)(((size_t)((n0 >> 6) + (p >> 2)) * nk) * 4096 + (p & 3) * 1024 + lane * 16);
                dstB[q] = A_BYTES + p * 1024 + lane * 16;
            }
            stepB = 4096;
        } else {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int chunk = w4 * NB + q, row = chunk * 8 + (lane >> 3), c = lane & 7;
                offB[q] = (unsigned)(((size_t)(n0 + row) * g.ldw + c * 8) * 2);
                dstB[q] = A_BYTES + row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
            }
            stepB = BK * 2;
        }
        u32x4 r0[GR], r1[GR];
        // The loads are inline asm: hipcc's own vmcnt bookkeeping merges the two register batches at the loop
        // head and drains BOTH before the first ds_write (prefetch depth 1); here the wait is counted by hand
        // -- loads return in order, so "at most GR outstanding" means the older batch has landed.
        auto fetch = [&](int kt, u32x4 (&r)[GR]) {
            if (QV_ABL(8)) return;
            const int sa = kt * (BK * 2), sb = kt * stepB;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r[q]) : "v"(offA[q]), "s"(rsA), "s"(sa) : "memory");
#pragma unroll
            for (int q = 0; q < NB; ++q)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r[4 + q]) : "v"(offB[q]), "s"(rsB), "s"(sb) : "memory");
        };
        auto put = [&](int kt, const u32x4 (&r)[GR], bool newer_in_flight) {
            if (newer_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GR) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned char *st = smem + (kt % NST) * STAGE_BYTES;
            if (QV_ABL(4)) return;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(u32x4 *)(st + dstA[q]) = r[q];
#pragma unroll
            for (int q = 0; q < NB; ++q) *(u32x4 *)(st + dstB[q]) = r[4 + q];
        };
        // Barrier kt: stage kt is written (loaders) and stage kt - 1 is read (consumers).  Between barriers
        // kt - 1 and kt a loader writes stage kt -- the buffer the consumers left before barrier kt - 1 --
        // out of registers that were requested two K-steps ago, then requests K-step kt + 2 into them.
        // put + fetch fused: write piece q of K-step kt, re-request it for K-step kt + 2 -- alternating stores and
        // loads keeps the LDS store path and the address path busy at the same time
        auto swap = [&](int kt, u32x4 (&r)[GR]) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GR) : "memory");   // (a newer batch is always in flight here)
            unsigned char *st = smem + (kt % NST) * STAGE_BYTES;
            const int sa = (kt + 2) * (BK * 2), sb = (kt + 2) * stepB;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!QV_ABL(4)) *(u32x4 *)(st + dstA[q]) = r[q];
                if (!QV_ABL(8)) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r[q]) : "v"(offA[q]), "s"(rsA), "s"(sa) : "memory");
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                if (!QV_ABL(4)) *(u32x4 *)(st + dstB[q]) = r[4 + q];
                if (!QV_ABL(8)) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r[4 + q]) : "v"(offB[q]), "s"(rsB), "s"(sb) : "memory");
            }
        };
        fetch(0, r0);
        if (nk > 1) fetch(1, r1);
        for (int kt = 0; kt < nk; kt += 2) {
            int kt_ = kt;
            QV_TRACE(0);                       // barrier kt - 1 released
            if (kt + 2 < nk && QV_SWAP) swap(kt, r0);
            else { put(kt, r0, kt + 1 < nk); if (kt + 2 < nk) fetch(kt + 2, r0); }
            QV_TRACE(1);                       // (s_memtime waits lgkmcnt: loads landed + writes done)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            QV_TRACE(2);                       // arriving at barrier kt
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < nk) {
                kt_ = kt + 1;
                QV_TRACE(0);
                if (kt + 3 < nk && QV_SWAP) swap(kt + 1, r1);
                else { put(kt + 1, r1, kt + 2 < nk); if (kt + 3 < nk) fetch(kt + 3, r1); }
                QV_TRACE(1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                QV_TRACE(2);
                __builtin_amdgcn_s_barrier();
            }
        }
    } else if (loader) {
#pragma unroll
        for (int s = 0; s < NST - 1; ++s)
            if (s < nk) stage(s);
        for (int kt = 0; kt < nk; ++kt) {
            const int ahead = min(nk - 1 - kt, NST - 2);  // stages issued after stage kt
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + NST - 1 < nk) stage(kt + NST - 1);
        }
    }
    for (int kt = 0; kt < (loader ? 0 : nk); ++kt) {
        const int kt_ = kt;
        QV_TRACE(2);                           // arriving at barrier kt (K-step kt - 1 computed)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt == 0) QV_PHASE(1);
        QV_TRACE(0);                           // barrier kt released
        const half_t *sA = (const half_t *)(smem + (kt % NST) * STAGE_BYTES);
        const half_t *sB = (const half_t *)((const unsigned char *)sA + A_BYTES);
        half2_t sc[NF], zo[NF];
        if (W4) {
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                half2_t sz = sS[(kt >> 1) * BN + wn * WN + j * 32 + (lane & 31)];
                sc[j] = half2_t{sz[0], sz[0]};
                zo[j] = half2_t{sz[1], sz[1]};
            }
        }
        // Fragment reads are software-pipelined over the four 16-deep sub-steps: the reads of sub-step ks + 1
        // are in flight while the MFMAs of sub-step ks run (two register sets).  With one set the next reads
        // could only issue behind the last MFMA of a sub-step and every sub-step exposed one LDS latency
        // (~90 of ~220 cycles, tools/gemm_trace.hip: 964 cycles per K-step for 512 cycles of MFMA).
        struct Raw { half8 a[2]; half8 b[NF]; uint32_t q4[NF]; uint2 q8[NF]; };
        auto rd = [&](int ks, Raw &f) {
            if (QV_ABL(2)) return;
            const int c = ks * 2 + (lane >> 5);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + (lane & 31);
                f.a[i] = *(const half8 *)(sA + row * 64 + ((c ^ ((row >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int row = wn * WN + j * 32 + (lane & 31);
                if (W4) {
                    // 32-byte rows, 4-byte chunks at c ^ ((row >> 2) & 7): the 64 lanes of the
                    // ds_read_b32 hit 64 different banks
                    f.q4[j] = *(const uint32_t *)((const unsigned char *)sB + row * 32 + ((c ^ ((row >> 2) & 7)) << 2));
                } else if (W8) {
                    // 64-byte rows, 8-byte chunks at c ^ ((row >> 2) & 7); bytes are q + 128
                    f.q8[j] = *(const uint2 *)((const unsigned char *)sB + (row >> 6) * 4096 + (row & 63) * 64 +
                                               ((c ^ ((row >> 2) & 7)) << 3));
                } else {
                    f.b[j] = *(const half8 *)(sB + row * 64 + ((c ^ ((row >> 1) & 7)) << 3));
                }
            }
        };
        auto mma = [&](const Raw &f) {
            if (QV_ABL(1)) {
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(f.a[i]));
#pragma unroll
                for (int j = 0; j < NF; ++j) { asm volatile("" ::"v"(f.b[j])); asm volatile("" ::"v"(f.q4[j])); asm volatile("" ::"v"(f.q8[j])); }
                return;
            }
            half8 b[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) b[j] = W4 ? dequant8(f.q4[j], sc[j], zo[j]) : W8 ? dequant8_i8(f.q8[j]) : f.b[j];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    // operands swapped (D^T = W A^T): a lane holds 4 CONSECUTIVE output columns per
                    // register quad -> 8/16-byte LDS writes in the epilogue
                    if constexpr (I8)
                        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, b[j]), __builtin_bit_cast(i32x4, f.a[i]),
                                                                          acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], f.a[i], acc[i][j], 0, 0, 0);
                }
        };
        // (sched_barrier: left alone, the scheduler sinks every read group back behind the previous MFMAs
        // to save the 16 registers)
        Raw f0 = {}, f1 = {};
        rd(0, f0);
        rd(1, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(f0);
        __builtin_amdgcn_sched_barrier(0);
        rd(2, f0);
        __builtin_amdgcn_sched_barrier(0);
        mma(f1);
        __builtin_amdgcn_sched_barrier(0);
        rd(3, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(f0);
        __builtin_amdgcn_sched_barrier(0);
        mma(f1);
    }
    __syncthreads();  // every wave is done with the operand stages before the epilogue reuses them
    QV_PHASE(2);
#if defined(QV_GEMM_TRACE) && defined(__HIP_DEVICE_COMPILE__)
    if (QV_ABL(32)) {   // no epilogue at all (accumulators kept live)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
#endif

    // ------------------------------------------------------------------ epilogue ----------
    // accumulator (i, j), register r: tile row  = wm*64 + i*32 + (lane & 31),
    //   tile column = wn*WN + j*32 + 8*(r >> 2) + 4*(lane >> 5) + (r & 3)
    const int l31 = lane & 31, hi = lane >> 5;
    // Every global access of the epilogue goes through MUBUF (SGPR descriptor + 32-bit byte offset): while the
    // co-resident block's consumer waves keep the SIMDs' matrix pipes busy, FLAT-encoded loads and stores of
    // this block would wait for a gap in their MFMA stream (see the LD = 1 note above).
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc((void *)g.bias, 0, g.N * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_scl = __builtin_amdgcn_make_buffer_rsrc((void *)(W8 ? g.w8scale : g.bias), 0, g.N * 4, 0x00020000);
    auto ldf4 = [&](const __amdgpu_buffer_rsrc_t &rs, int elem) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, elem * 4, 0, 0));
    };

    if constexpr (I8) {
        // ---- A8W8 epilogue (qv_ort.h): int32 accumulator -> float32, per-utterance activation scale, bias, activation.
        //   v = float(acc + (128 - zp_x[u]) * wsum[n]) * (s_x[u] * s_w) + bias[n]
        // The int32 sum is < 2^24 in magnitude (K <= 512 products of |255| x |127|), so the conversion is exact and v
        // carries exactly the two roundings the reference's Cast -> Mul -> Add chain has.
        constexpr bool GLU = EPI == EPI_GLU, OUT16 = EPI == EPI_F16_RELU;
        constexpr bool RELU = EPI == EPI_F16_RELU || EPI == EPI_F32_RELU;
        constexpr bool FOLD = EPI == EPI_GLU || EPI == EPI_F32_RELU;   // the output feeds another quantiser: track its range
        constexpr int BNO = GLU ? BN / 2 : BN;
        constexpr int LDT = BNO + 4;                                   // floats per staged row
        float *sO = (float *)smem;
        uint32_t *sMM = (uint32_t *)(smem + BM * LDT * 4);             // [BM][2] range keys of the tile's rows
        auto row_owner = [&](int grow, bool &valid) {
            valid = true;
            if (g.row_map) return g.row_map[grow] >> 16;
            const int u = grow / g.rows_per_utt;
            valid = (grow - u * g.rows_per_utt) / g.f_per_t < g.len[u];
            return u;
        };
        if (FOLD) {
            for (int k = tid; k < BM * 2; k += NT) sMM[k] = (k & 1) ? 0u : 0xFFFFFFFFu;   // empty range
            __syncthreads();
        }
        if (!loader) {
            const __amdgpu_buffer_rsrc_t rs_ws = __builtin_amdgcn_make_buffer_rsrc((void *)g.wsum, 0, g.N * 4, 0x00020000);
            float srow[2];
            int zc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int grow = m0 + wm * 64 + i * 32 + l31;
                grow = grow < g.M ? grow : g.M - 1;
                bool valid;
                const QParam p = dql_param(g.mm_in + QV_MM_STRIDE * row_owner(grow, valid));
                srow[i] = p.scale * g.w_scale;
                zc[i] = 128 - (int)p.zp;
            }
            float mn[2] = {INFINITY, INFINITY}, mx[2] = {-INFINITY, -INFINITY};
            if (GLU) {
                // W rows interleaved in 32-channel groups: [value(32) | gate(32)] per 64 columns
                const int nb = n0 + wn * WN;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nl = 8 * q + 4 * hi;
                    const f32x4 ba = ldf4(rs_bias, nb + nl), bg = ldf4(rs_bias, nb + 32 + nl);
                    const i32x4 wa = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ws, (nb + nl) * 4, 0, 0));
                    const i32x4 wg = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ws, (nb + 32 + nl) * 4, 0, 0));
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int rl = wm * 64 + i * 32 + l31;
                        f32x4 o, av, gv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            av[e] = (float)(acc[i][0][q * 4 + e] + zc[i] * wa[e]) * srow[i] + ba[e];
                            gv[e] = (float)(acc[i][NF - 1][q * 4 + e] + zc[i] * wg[e]) * srow[i] + bg[e];
                        }
                        const f32x4 sg = sigmoid4(gv);   // (v_exp_f32 / v_rcp_f32: a few 1e-7 relative, like any libm's expf)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[e] = av[e] * sg[e];
                            mn[i] = fminf(mn[i], o[e]);
                            mx[i] = fmaxf(mx[i], o[e]);
                        }
                        *(f32x4 *)(sO + rl * LDT + wn * (WN / 2) + nl) = o;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NF; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cl = wn * WN + j * 32 + 8 * q + 4 * hi;
                        const f32x4 bb = ldf4(rs_bias, n0 + cl);
                        const i32x4 ws = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ws, (n0 + cl) * 4, 0, 0));
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int rl = wm * 64 + i * 32 + l31;
                            f32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float v = (float)(acc[i][j][q * 4 + e] + zc[i] * ws[e]) * srow[i] + bb[e];
                                if (RELU) v = v > 0.f ? v : 0.f;
                                o[e] = v;
                                mn[i] = fminf(mn[i], v);
                                mx[i] = fmaxf(mx[i], v);
                            }
                            *(f32x4 *)(sO + rl * LDT + cl) = o;
                        }
                    }
            }
            if (FOLD) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int rl = wm * 64 + i * 32 + l31;
                    atomicMin(&sMM[2 * rl], fenc(mn[i]));
                    atomicMax(&sMM[2 * rl + 1], fenc(mx[i]));
                }
            }
        }
        __syncthreads();
        if (FOLD) {
            // One atomic pair per (tile, utterance), not per row: all tiles start at once and every row's pair would
            // land on its utterance's two addresses (measured: 200 us for this GEMM instead of 15).  A tile's rows are
            // sorted by utterance, so the first row of each run reduces the run.
            int *sU = (int *)(sMM + 2 * BM);       // [BM] owner of each row, -1 = not a valid row
            if (tid < BM) {
                bool valid = false;
                int u = -1;
                if (m0 + tid < g.M) u = row_owner(m0 + tid, valid);
                sU[tid] = valid && sMM[2 * tid] <= sMM[2 * tid + 1] ? u : -1;
            }
            __syncthreads();
            if (tid < BM && sU[tid] >= 0 && (tid == 0 || sU[tid - 1] != sU[tid])) {
                uint32_t kn = sMM[2 * tid], kx = sMM[2 * tid + 1];
                for (int k = tid + 1; k < BM && sU[k] == sU[tid]; ++k) { kn = min(kn, sMM[2 * k]); kx = max(kx, sMM[2 * k + 1]); }
                mm_fold_keys(g.mm_out + QV_MM_STRIDE * sU[tid], kn, kx);
            }
        }
        // the output descriptor is rebased on this tile's first row: the dense subsampling tensors have millions of rows
        // (4-byte elements), and a whole-tensor descriptor would need more than its 32-bit size / offsets hold
        const int tile_rows = g.M - m0 < BM ? g.M - m0 : BM;
        if (OUT16) {
            constexpr int CPR = BNO / 8;   // 16-byte chunks of halves per row
            const __amdgpu_buffer_rsrc_t rs_out16 = __builtin_amdgcn_make_buffer_rsrc((half_t *)g.out + (size_t)m0 * g.ldo, 0,
                                                                                      tile_rows * g.ldo * 2, 0x00020000);
            for (int idx = tid; idx < BM * CPR; idx += NT) {
                const int r = idx / CPR, c = (idx % CPR) * 8;
                if (r >= tile_rows) continue;
                const f32x4 a = *(const f32x4 *)(sO + r * LDT + c), b = *(const f32x4 *)(sO + r * LDT + c + 4);
                const half8 h = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, h), rs_out16, (r * g.ldo + n0 + c) * 2, 0, 0);
            }
        } else {
            constexpr int CPR = BNO / 4;            // 16-byte chunks per row
            constexpr int IT = BM * CPR / NT;       // chunks per thread (exact)
            const int n0o = GLU ? n0 / 2 : n0;
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((float *)g.out + (size_t)m0 * g.ldo, 0,
                                                                                    tile_rows * g.ldo * 4, 0x00020000);
            f32x4 old[IT];
            if (EPI == EPI_RESID) {
#pragma unroll
                for (int k = 0; k < IT; ++k) {
                    const int idx = tid + k * NT, r = idx / CPR, c = (idx % CPR) * 4;
                    if (r < tile_rows) old[k] = ldf4(rs_out, r * g.ldo + n0o + c);
                }
            }
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int idx = tid + k * NT, r = idx / CPR, c = (idx % CPR) * 4;
                if (r >= tile_rows) continue;
                f32x4 v = *(const f32x4 *)(sO + r * LDT + c);
                if (EPI == EPI_RESID) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = old[k][e] + g.alpha * v[e];
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs_out, (r * g.ldo + n0o + c) * 4, 0, 0);
            }
        }
        return;
    }

    if (EPI == EPI_QKV && n0 >= 2 * QV_D) {
        // V tile: stored TRANSPOSED, Vt[b][h*64+d][t].  The tile goes through LDS as [d][frame] so
        // that all 8 waves write it out with the 64 lanes of a store on 128 consecutive frames of
        // one d (4-byte pairs, 256 B per instruction); stores straight from the accumulator layout
        // (2-byte elements, 32 frames per d) cost 8 us per GEMM.
        constexpr int LDV = BM + 2;   // halves per d row (odd dword pitch: d rows rotate over the banks)
        half_t *sT = (half_t *)smem;
        if (!loader) {
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = wn * WN + j * 32 + 8 * q + 4 * hi;
                    const f32x4 bb = ldf4(rs_bias, n0 + cl);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int rl = wm * 64 + i * 32 + l31;
#pragma unroll
                        for (int e = 0; e < 4; ++e) sT[(cl + e) * LDV + rl] = (half_t)(acc[i][j][q * 4 + e] + bb[e]);
                    }
                }
        }
        __syncthreads();
        // thread -> frame pair (fixed), then d = tid / 64 + 8 k
        const int r0 = m0 + 2 * lane, r1 = r0 + 1;
        const int bt0 = r0 < g.M ? g.row_map[r0] : -1, bt1 = r1 < g.M ? g.row_map[r1] : -1;
        const bool pair = bt0 >= 0 && bt1 == bt0 + 1 && (bt0 & 1) == 0;   // same utterance, even frame: one 4-byte store
        half_t *vt = (half_t *)g.out2;
        for (int d = wave; d < BN; d += NT / 64) {
            const size_t drow = (size_t)(n0 - 2 * QV_D + d);
            const half_t v0 = sT[d * LDV + 2 * lane], v1 = sT[d * LDV + 2 * lane + 1];
            if (pair) {
                half2_t v = {v0, v1};
                *(half2_t *)(vt + ((size_t)(bt0 >> 16) * QV_D + drow) * g.t_pad + (bt0 & 0xFFFF)) = v;
            } else {
                if (bt0 >= 0) vt[((size_t)(bt0 >> 16) * QV_D + drow) * g.t_pad + (bt0 & 0xFFFF)] = v0;
                if (bt1 >= 0) vt[((size_t)(bt1 >> 16) * QV_D + drow) * g.t_pad + (bt1 & 0xFFFF)] = v1;
            }
        }
        QV_PHASE(3);
        return;
    }

    // this wave's bias values, requested together (g.bias is never null: one load latency, not
    // one per register quad)
    f32x4 bia[NF][4], scl[NF][4];   // scl: per-channel weight scales (W8A16 only)
    if (EPI != EPI_GLU && !loader) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bia[j][q] = ldf4(rs_bias, n0 + wn * WN + j * 32 + 8 * q + 4 * hi);
                if (W8) scl[j][q] = ldf4(rs_scl, n0 + wn * WN + j * 32 + 8 * q + 4 * hi);
            }
    }

    if (epi_is_f32(EPI)) {
        constexpr int LDT = BN + 4;  // floats per staged row (pad keeps 16-B alignment)
        float *sO = (float *)smem;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (loader) continue;   // only the consumer waves hold accumulators
            const int rl = wm * 64 + i * 32 + l31;
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = wn * WN + j * 32 + 8 * q + 4 * hi;
                    f32x4 v;
                    const f32x4 bb = bia[j][q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[i][j][q * 4 + e];
                        if (W8) x *= scl[j][q][e];
                        v[e] = g.alpha * (x + bb[e]);
                    }
                    *(f32x4 *)(sO + rl * LDT + cl) = v;
                }
        }
        __syncthreads();
        constexpr int CPR = BN / 4;            // 16-byte chunks per row
        constexpr int IT = BM * CPR / NT;      // chunks per thread (exact)
        // residual: all of a thread's old values are requested before the first store (a store
        // followed by the next load of the same array would serialise one latency per chunk)
        f32x4 old[IT];
        // (the descriptor's size bounds the rows: loads past row M return 0, stores past it are dropped)
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)((size_t)g.M * g.ldo * 4), 0x00020000);
        if (EPI == EPI_RESID) {
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                int idx = tid + k * NT, r = idx / CPR, c = (idx % CPR) * 4;
                if (m0 + r < g.M && !QV_ABL(16)) old[k] = ldf4(rs_out, (m0 + r) * g.ldo + n0 + c);
            }
        }
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            int idx = tid + k * NT, r = idx / CPR, c = (idx % CPR) * 4;
            if (m0 + r >= g.M) continue;
            f32x4 v = *(const f32x4 *)(sO + r * LDT + c);
            if (EPI == EPI_RESID) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += old[k][e];
            }
            if (QV_ABL(16)) { asm volatile("" ::"v"(v)); continue; }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs_out, ((m0 + r) * g.ldo + n0 + c) * 4, 0, 0);
        }
        QV_PHASE(3);
        return;
    }

    {
        constexpr int BNO = EPI == EPI_GLU ? BN / 2 : BN;  // output tile width
        constexpr int LDT = BNO + 8;                        // halves per staged row
        half_t *sO = (half_t *)smem;
        const int n0o = EPI == EPI_GLU ? n0 / 2 : n0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (loader) continue;   // only the consumer waves hold accumulators
            const int rl = wm * 64 + i * 32 + l31;
            if (EPI == EPI_GLU) {
                // W rows interleaved in 32-channel groups: [value(32) | gate(32)] per 64 columns
                const int nb = n0 + wn * WN;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int nl = 8 * q + 4 * hi;
                    f32x4 ba = ldf4(rs_bias, nb + nl), bg = ldf4(rs_bias, nb + 32 + nl);
                    f32x4 sa = {1.f, 1.f, 1.f, 1.f}, sg = sa;
                    if (W8) { sa = ldf4(rs_scl, nb + nl); sg = ldf4(rs_scl, nb + 32 + nl); }
                    half4 o;
                    f32x4 av, gv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        av[e] = acc[i][0][q * 4 + e];
                        gv[e] = acc[i][NF - 1][q * 4 + e];
                        if (W8) { av[e] *= sa[e]; gv[e] *= sg[e]; }
                        av[e] += ba[e];
                        gv[e] += bg[e];
                    }
                    const f32x4 sgm = sigmoid4(gv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (half_t)(av[e] * sgm[e]);
                    *(half4 *)(sO + rl * LDT + wn * (WN / 2) + nl) = o;
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = wn * WN + j * 32 + 8 * q + 4 * hi;
                    const f32x4 bb = bia[j][q];
                    half4 o;
                    f32x4 xv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[e] = acc[i][j][q * 4 + e] + bb[e];
                    if (EPI == EPI_F16_SWISH) xv = swish4(xv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = xv[e];
                        if (EPI == EPI_F16_RELU) x = x > 0.f ? x : 0.f;
                        o[e] = (half_t)x;
                    }
                    *(half4 *)(sO + rl * LDT + cl) = o;
                }
        }
        __syncthreads();
        constexpr int CPR = BNO / 8;  // 16-byte chunks per row
        const __amdgpu_buffer_rsrc_t rs_out16 = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)((size_t)g.M * g.ldo * 2), 0x00020000);
        for (int idx = tid; idx < BM * CPR; idx += NT) {
            int r = idx / CPR, c = (idx % CPR) * 8;
            if (m0 + r >= g.M) continue;
            if (QV_ABL(16)) { asm volatile("" ::"v"(*(const half8 *)(sO + r * LDT + c))); continue; }
            __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4_t *)(sO + r * LDT + c), rs_out16, ((m0 + r) * g.ldo + n0o + c) * 2, 0, 0);
        }
        QV_PHASE(3);
    }
}

template <int EPI, int BN, int WQ, int NST, int LD = 0>
static void launch_one(const GemmArgs &g, hipStream_t s) {
    constexpr bool W4 = WQ == 4, W8 = WQ == 8;
    dim3 grid(g.N / BN, (g.M + 127) / 128);
    size_t lds = W4 ? NST * ((128 * 64 * 2) + (BN * 32)) + (size_t)BN * (g.K / 128) * 4
                    : W8 ? NST * ((128 * 64 * 2) + (BN * 64)) : NST * ((128 * 64 * 2) + (BN * 64 * 2));
    size_t epi = epi_is_f32(EPI) ? (size_t)128 * (BN + 4) * 4 : (size_t)128 * (BN + 8) * 2;
    if (WQ == 88) epi = (size_t)128 * ((EPI == EPI_GLU ? BN / 2 : BN) + 4) * 4 + 128 * 3 * 4;   // f32 staging + per-row range keys and owners
    if (epi > lds) lds = epi;
    // more than 64 KB of dynamic LDS is opted into, once per instantiation and only where needed
    if (lds > 64 * 1024) {
        static size_t allowed = 0;
        if (lds > allowed) {
            (void)hipFuncSetAttribute((const void *)k_gemm<EPI, BN, WQ, NST, LD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            allowed = lds;
        }
    }
    hipLaunchKernelGGL((k_gemm<EPI, BN, WQ, NST, LD>), grid, dim3(512), lds, s, g);
}

// tile width BN in {64, 128} and LDS stage count NST in {2, 3, 4} (see launch_gemm)
template <int EPI, int WQ>
static void launch_shape(const GemmArgs &g, hipStream_t s, bool narrow, int nst) {
    if constexpr (WQ == 8) {
        // int8 weights (the two pointwise-convolution shapes) only exist with register-staged loaders: their direct-to-LDS
        // variants (QVERSE_GEMM_LD=0) needed 4-12 spilled VGPRs at the 128-register budget, i.e. scratch traffic inside a
        // loop whose vmcnt waits are counted by hand (tests/test_capi_load.py keeps the binary free of such kernels)
        launch_one<EPI, 128, 8, 2, 1>(g, s);
    } else if (nst == 0) {   // register-staged loaders (2-stage LDS ring)
        if (narrow) launch_one<EPI, 64, WQ, 2, 1>(g, s);
        else launch_one<EPI, 128, WQ, 2, 1>(g, s);
    } else if (narrow) {
        if (nst == 2) launch_one<EPI, 64, WQ, 2>(g, s);
        else launch_one<EPI, 64, WQ, 3>(g, s);
    } else {
        if (nst == 2) launch_one<EPI, 128, WQ, 2>(g, s);
        else if (nst == 3) launch_one<EPI, 128, WQ, 3>(g, s);
        else launch_one<EPI, 128, WQ, 4>(g, s);
    }
}

// Symmetric block-128 int4: per row and per 128 consecutive k, scale = v / -8 where v is the
// element of largest magnitude (first one on ties), q = clamp(floor(w / scale + 8.5), 0, 15),
// w' = (q - 8) * half(scale).  oracle/fastconformer_ref.py:quant_dequant_int4 mirrors this exactly.
void qv_pack_w4(const float *w, int N, int K, uint8_t *q_out, half_t *scale_out) {
    const int nk = K / 64, nkb = K / 128;
    for (int n = 0; n < N; ++n) {
        const float *row = w + (size_t)n * K;
        for (int kb = 0; kb < nkb; ++kb) {
            float vmax = 0.f, amax = -1.f;
            for (int k = 0; k < 128; ++k) {
                float a = fabsf(row[kb * 128 + k]);
                if (a > amax) { amax = a; vmax = row[kb * 128 + k]; }
            }
            float scale = vmax / -8.0f;
            float rs = scale != 0.f ? 1.0f / scale : 0.f;
            scale_out[((size_t)kb * N + n) * 2] = (half_t)scale;
            scale_out[((size_t)kb * N + n) * 2 + 1] = (half_t)(1024.f + 8.f);  // symmetric: zero point 8
            for (int k = 0; k < 128; ++k) {
                float t = row[kb * 128 + k] * rs + 8.0f;
                int q = (int)floorf(t + 0.5f);
                q = q < 0 ? 0 : q > 15 ? 15 : q;
                int kk = kb * 128 + k, kt = kk >> 6, c = (kk & 63) >> 3, e = kk & 7;
                int nib = (e >> 1) + 4 * (e & 1);  // k = 2p -> nibble p, k = 2p + 1 -> nibble p + 4
                int pos = c ^ ((n >> 2) & 7);
                size_t byte = ((size_t)(n >> 6) * nk + kt) * 2048 + (size_t)(n & 63) * 32 + pos * 4 + (nib >> 1);
                if (nib & 1) q_out[byte] = (uint8_t)((q_out[byte] & 0x0F) | (q << 4));
                else q_out[byte] = (uint8_t)((q_out[byte] & 0xF0) | q);
            }
        }
    }
}

// The same device layout from a grid that is GIVEN (a weight file converted from the reference's quantised ONNX carries
// MatMulNBits' own block scales and zero points, tools/convert_weights.py): w[n][k] = (q - zp[n][kb]) * scale[n][kb] holds
// exactly in float32, so q = rint(w / scale + zp) returns the file's integer verbatim.  The device then multiplies by
// half(scale) -- the file's float32 scale rounded once, 2^-11 = 4.9e-4 relative at most -- and subtracts the file's zero point
// (asymmetric blocks included: the {scale, 1024 + zp} pair format always carried one).  Returns the number of elements
// that cannot be represented this way (0 for a consistent file): a recovered q that is not an integer in [0, 15] to 1e-3,
// and every element of a block whose zero point is not an integer in [0, 15] (half(1024 + zp) would round it) or whose
// scale leaves the normal range of f16 (it would be flushed or overflow on the device).  The caller keeps the dequantised
// f16 path for such a tensor (tools/convert_weights.py) instead of running it on a grid the file does not have.
int64_t qv_pack_w4_given(const float *w, int N, int K, const float *scale, const float *zp, uint8_t *q_out, half_t *scale_out) {
    const int nk = K / 64, nkb = K / 128;
    int64_t bad = 0;
    for (int n = 0; n < N; ++n) {
        const float *row = w + (size_t)n * K;
        for (int kb = 0; kb < nkb; ++kb) {
            const float sc = scale[(size_t)n * nkb + kb], z = zp[(size_t)n * nkb + kb];
            scale_out[((size_t)kb * N + n) * 2] = (half_t)sc;
            scale_out[((size_t)kb * N + n) * 2 + 1] = (half_t)(1024.f + z);
            const float asc = fabsf(sc);
            if (z != nearbyintf(z) || z < 0.f || z > 15.f || (sc != 0.f && (asc < 6.103515625e-05f || asc > 65504.f))) bad += 128;
            for (int k = 0; k < 128; ++k) {
                const float t = sc != 0.f ? row[kb * 128 + k] / sc + z : z;
                int q = (int)nearbyintf(t);
                if (fabsf(t - (float)q) > 1e-3f || q < 0 || q > 15) ++bad;
                q = q < 0 ? 0 : q > 15 ? 15 : q;
                int kk = kb * 128 + k, kt = kk >> 6, c = (kk & 63) >> 3, e = kk & 7;
                int nib = (e >> 1) + 4 * (e & 1);
                int pos = c ^ ((n >> 2) & 7);
                size_t byte = ((size_t)(n >> 6) * nk + kt) * 2048 + (size_t)(n & 63) * 32 + pos * 4 + (nib >> 1);
                if (nib & 1) q_out[byte] = (uint8_t)((q_out[byte] & 0x0F) | (q << 4));
                else q_out[byte] = (uint8_t)((q_out[byte] & 0xF0) | q);
            }
        }
    }
    return bad;
}

// Per-row symmetric int8: scale = max|w| / 127 (1 for an all-zero row), q = clamp(floor(w / scale + 0.5), -127, 127),
// stored as q + 128.  oracle/fastconformer_ref.py:quant_dequant_int8 mirrors this exactly.
void qv_pack_w8(const float *w, int N, int K, uint8_t *q_out, float *scale_out) {
    const int nk = K / 64;
    for (int n = 0; n < N; ++n) {
        const float *row = w + (size_t)n * K;
        float amax = 0.f;
        for (int k = 0; k < K; ++k) amax = fmaxf(amax, fabsf(row[k]));
        const float scale = amax > 0.f ? amax / 127.0f : 1.0f;
        scale_out[n] = scale;
        for (int k = 0; k < K; ++k) {
            float t = row[k] / scale;
            int q = (int)floorf(t + 0.5f);
            q = q < -127 ? -127 : q > 127 ? 127 : q;
            const int kt = k >> 6, c = (k & 63) >> 3;
            const size_t byte = ((size_t)(n >> 6) * nk + kt) * 4096 + (size_t)(n & 63) * 64 + ((c ^ ((n >> 2) & 7)) << 3) + (k & 7);
            q_out[byte] = (uint8_t)(q + 128);
        }
    }
}

// ---- measurement hook: HIP-event timing of every GEMM launch, per (epilogue, tile) class ----
namespace {
struct GemmProf {
    bool on = false;
    std::vector<hipEvent_t> ev;   // pairs
    std::vector<int> cls;
    std::vector<double> flops;
} g_prof;
}  // namespace

bool qv_gemm_prof_on() { return g_prof.on; }

void qv_gemm_prof_enable(bool on) {
    g_prof.on = on;
    g_prof.cls.clear();
    g_prof.flops.clear();
}

// call after the stream has been synchronised; accumulates into ms[21], flops[21], n[21]
// (class = epilogue * 3 + tile: 0 = 64-wide, 1 = 128-wide, 2 = 256 x 256)
void qv_gemm_prof_collect(double *ms, double *flops, int *n) {
    for (int i = 0; i < 21; ++i) { ms[i] = 0; flops[i] = 0; n[i] = 0; }
    for (size_t i = 0; i < g_prof.cls.size(); ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) continue;
        ms[g_prof.cls[i]] += t;
        flops[g_prof.cls[i]] += g_prof.flops[i];
        n[g_prof.cls[i]] += 1;
    }
    g_prof.cls.clear();
    g_prof.flops.clear();
}

static void launch_gemm_inner(int epi, const GemmArgs &g, hipStream_t s, bool narrow, int nst) {
    if (g.Wi8) {
        // int8 activations x int8 weights (QV_PREC_ORT_MIXED): the GEMM-shaped convolutions; 128-wide tiles,
        // register-staged loaders
        if (g.N % 128 != 0 || !g.wsum || !g.mm_in) abort();
        switch (epi) {
            case EPI_GLU: launch_one<EPI_GLU, 128, 88, 2, 1>(g, s); break;
            case EPI_RESID: launch_one<EPI_RESID, 128, 88, 2, 1>(g, s); break;
            case EPI_F32: launch_one<EPI_F32, 128, 88, 2, 1>(g, s); break;
            case EPI_F32_RELU: launch_one<EPI_F32_RELU, 128, 88, 2, 1>(g, s); break;
            case EPI_F16_RELU: launch_one<EPI_F16_RELU, 128, 88, 2, 1>(g, s); break;
            default: abort();
        }
        return;
    }
    if (g.Wq) {
        // int4 weights: the Linear layers only (FFN, QKV, attention out, linear_pos)
        switch (epi) {
            case EPI_F16: launch_shape<EPI_F16, 4>(g, s, narrow, nst); break;
            case EPI_F16_SWISH: launch_shape<EPI_F16_SWISH, 4>(g, s, narrow, nst); break;
            case EPI_RESID: launch_shape<EPI_RESID, 4>(g, s, narrow, nst); break;
            case EPI_QKV: launch_shape<EPI_QKV, 4>(g, s, narrow, nst); break;
            default: abort();
        }
        return;
    }
    if (g.W8) {
        // int8 weights: the pointwise convolutions of the conv module (GLU and residual epilogues)
        if (narrow || g.N % 128 != 0) abort();
        switch (epi) {
            case EPI_GLU: launch_shape<EPI_GLU, 8>(g, s, false, nst > 3 ? 3 : nst); break;
            case EPI_RESID: launch_shape<EPI_RESID, 8>(g, s, false, nst > 3 ? 3 : nst); break;
            default: abort();
        }
        return;
    }
    switch (epi) {
        case EPI_F16: launch_shape<EPI_F16, 0>(g, s, narrow, nst); break;
        case EPI_F16_SWISH: launch_shape<EPI_F16_SWISH, 0>(g, s, narrow, nst); break;
        case EPI_F16_RELU: launch_shape<EPI_F16_RELU, 0>(g, s, narrow, nst); break;
        case EPI_RESID: launch_shape<EPI_RESID, 0>(g, s, narrow, nst); break;
        case EPI_F32: launch_shape<EPI_F32, 0>(g, s, narrow, nst); break;
        case EPI_QKV: launch_shape<EPI_QKV, 0>(g, s, narrow, nst); break;
        case EPI_GLU: launch_shape<EPI_GLU, 0>(g, s, false, nst); break;
        default: abort();
    }
}

static std::atomic<int> g_t256{-1}, g_bm{-1}, g_policy_epoch{0};
void qv_gemm_set_t256(int mode) { g_t256.store(mode); g_policy_epoch.fetch_add(1); }
void qv_gemm_set_bm(int mode) { g_bm.store(mode); g_policy_epoch.fetch_add(1); }
int qv_gemm_policy_epoch() { return g_policy_epoch.load(); }

namespace {
struct GemmPlan { bool wide, narrow; int nst, bm; };

// Tile and pipeline choice (measured per shape, tools/gemm_bench.hip):
//  * 256 x 256 tiles, one block per CU (qv_gemm256.hip), when N % 256 == 0 and the grid is large enough (see below:
//    FFN-up and QKV at B = 64 x 10 s, every N >= 512 shape at B = 256; more shapes with >= 3 batches in flight);
//  * otherwise 128-wide tiles whenever N allows, register-staged loader waves (QVERSE_GEMM_LD=0: direct global->LDS
//    loads with 2 stages at >= 400 tiles (two 64 KB blocks per CU), else 3, 4 when the K loop is long);
//  * int4 weights with too few 128-wide tiles for two blocks per CU (FFN-down, out-projection: N = 512): the consumer
//    wave of a lone block pays its dequantisation VALU in front of its own MFMAs; 64-wide tiles give every SIMD a
//    second consumer wave to interleave with.
GemmPlan gemm_plan(int epi, const GemmArgs &g) {
    static const int env_nst = [] { const char *e = getenv("QVERSE_GEMM_NST"); return e ? atoi(e) : 0; }();
    static const int env_narrow = [] { const char *e = getenv("QVERSE_GEMM_NARROW"); return e ? atoi(e) : -1; }();
    static const int env_ld = [] { const char *e = getenv("QVERSE_GEMM_LD"); return e ? atoi(e) : 1; }();
    static const int env_t256 = [] { const char *e = getenv("QVERSE_GEMM_T256"); return e ? atoi(e) : 1; }();
    GemmPlan p;
    p.narrow = g.N % 128 != 0;
    if (g.Wq && !p.narrow && (g.N / 128) * ((g.M + 127) / 128) < 400) p.narrow = true;
    if (env_narrow >= 0 && g.N % 128 == 0 && epi != EPI_GLU) p.narrow = env_narrow != 0;
    const int tiles = (g.N / (p.narrow ? 64 : 128)) * ((g.M + 127) / 128);
    p.nst = tiles >= 400 ? 2 : (g.K / 64 >= 16 ? 4 : 3);
    if (p.narrow && p.nst > 3) p.nst = 3;
    if (env_nst >= 2 && env_nst <= (p.narrow ? 3 : 4)) p.nst = env_nst;
    if (env_ld == 1) p.nst = 0;
    const int t256v = g_t256.load(), t256 = t256v >= 0 ? t256v : env_t256;
    p.wide = false;
    p.bm = 256;
    if (t256 > 0 && g.N % 256 == 0 && g.bias && !(g.Wq && (g.K % 128 != 0 || g.K > 4096))) {
        const int tiles256 = (g.N / 256) * ((g.M + 255) / 256);
        // one batch at a time: a 256 x 256 grid must cover most of the chip (>= 160 tiles) or the 128-wide kernel's
        // 252+ blocks finish sooner; with >= 3 batches in flight the other batches' kernels take the idle CUs and
        // what counts is CU time per GEMM -- 128 tiles are enough, and the long-K shapes (FFN-down, the
        // subsampling projection: 64 tiles at B = 64 x 10 s) go wide as well (+2.3 % end to end, same-box sweep)
        static const int env_min = [] { const char *e = getenv("QVERSE_GEMM_MINTILES"); return e ? atoi(e) : 0; }();
        static const int env_min_longk = [] { const char *e = getenv("QVERSE_GEMM_MINTILES_LONGK"); return e ? atoi(e) : 0; }();
        const bool busy = g.in_flight >= 3;
        // (four or more batches in flight: wide tiles wherever the shape allows, +1.4 % over the 128-tile threshold)
        const int min_tiles = env_min > 0 ? env_min : g.in_flight >= 4 ? 1 : busy ? 128 : 160;
        const int min_tiles_longk = env_min_longk > 0 ? env_min_longk : busy ? 1 : 160;
        p.wide = t256 >= 2 || tiles256 >= (g.K >= 2048 ? min_tiles_longk : min_tiles);
        // Tile height (round 6).  A 192-row tile moves 17 % more operand bytes per flop, so it is only worth its rounds:
        // one batch at a time a GEMM takes ceil(tiles / 256 CUs) rounds of one tile time each (M = 24,064 = 64 clips x
        // 30 s, N = 512: 188 tiles of 256 rows = one round with 68 CUs idle, 252 tiles of 192 rows = one round of 3/4 the
        // length).  With three or more batches in flight the other batches' kernels take the idle CUs and CU time per
        // GEMM is what counts (mode 2 applies the rule there too; tools/gemm_bench + bench.py measure both).
        static const int env_bm = [] { const char *e = getenv("QVERSE_GEMM_BM"); return e ? atoi(e) : 1; }();
        const int bmv = g_bm.load(), bm_mode = bmv >= 0 ? bmv : env_bm;
        if (p.wide && bm_mode > 0) {
            const long tiles192 = (long)(g.N / 256) * ((g.M + 191) / 192);
            const long cost256 = ((tiles256 + 255) / 256) * 256, cost192 = ((tiles192 + 255) / 256) * 192;
            if (bm_mode >= 3 || ((bm_mode == 2 || !busy) && cost192 * 108 <= cost256 * 100)) p.bm = 192;
        }
    }
    return p;
}
}  // namespace

// "k_gemm256<f16_swish>" / "k_gemm<resid,128>": the kernel launch_gemm picks for this call (measurement reports)
const char *qv_gemm_kernel_name(int epi, const GemmArgs &g) {
    static const char *EPI[8] = {"f16", "f16_swish", "f16_relu", "glu", "resid", "f32", "qkv", "f32_relu"};
    static thread_local char buf[64];
    const GemmPlan p = gemm_plan(epi, g);
    if (g.Wi8) snprintf(buf, sizeof buf, p.wide ? (p.bm == 192 ? "k_gemm256<%s,a8w8,192>" : "k_gemm256<%s,a8w8>") : "k_gemm<%s,128,a8w8>", EPI[epi]);
    else if (p.wide) snprintf(buf, sizeof buf, p.bm == 192 ? "k_gemm256<%s,192>" : "k_gemm256<%s>", EPI[epi]);
    else snprintf(buf, sizeof buf, "k_gemm<%s,%d>", EPI[epi], p.narrow ? 64 : 128);
    return buf;
}

void launch_gemm(int epi, const GemmArgs &g, hipStream_t s) {
    if (g.K % 64 != 0 || g.N % 64 != 0 || (g.Wq && g.K % 128 != 0)) {
        fprintf(stderr, "launch_gemm: unsupported shape N=%d K=%d\n", g.N, g.K);
        abort();
    }
    const GemmPlan p = gemm_plan(epi, g);
    auto go = [&] {
        if (p.wide && launch_gemm256(epi, g, s, p.bm)) return;
        launch_gemm_inner(epi, g, s, p.narrow, p.nst);
    };
    if (!g_prof.on) { go(); return; }
    size_t i = g_prof.cls.size();
    while (g_prof.ev.size() < 2 * (i + 1)) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { go(); return; }
        g_prof.ev.push_back(e);
    }
    (void)hipEventRecord(g_prof.ev[2 * i], s);
    go();
    (void)hipEventRecord(g_prof.ev[2 * i + 1], s);
    g_prof.cls.push_back((epi == EPI_F32_RELU ? EPI_F16_RELU : epi) * 3 + (p.wide ? 2 : p.narrow ? 0 : 1));   // (21 classes in the ABI)
    g_prof.flops.push_back(2.0 * (double)g.M * (double)g.N * (double)g.K * (g.Wi8 ? 2.0 : 1.0));
}
