// qv_gemm256.hip -- the wide-tile variant of the fused-epilogue f16 GEMM (see qv_gemm.hip, qv_kernels.h).
//
// C[M,N] = epilogue(A[M,K] * W[N,K]^T + bias) on 256 x 256 x 64 tiles, ONE 512-thread block per CU.
//
// Why a second tile shape.  The 128 x 128 kernel (4 consumer waves of 64 x 64 + 4 loader waves, two blocks per CU)
// moves 32 KB through the CU's vector-memory path (64 B/clk) and 64 KB of fragment reads + 32 KB of stage
// writes through the LDS array (256 B/clk reads, 128 B/clk wide writes) per 512 matrix-pipe cycles: both
// paths are ~100 % busy at full MFMA rate, which is why its K loop sits at ~64 % matrix-pipe occupancy whatever
// the loaders do (DESIGN.md "GEMM, round 2").  A 256 x 256 tile halves the operand bytes per flop on both
// paths: per 64-deep K-tile a CU loads 64 KB (1024 cycles of the 64 B/clk path) and reads 192 KB of fragments
// + writes 64 KB of stages (~1280 LDS-array cycles) for 2048 matrix-pipe cycles.
//
// Structure: 8 waves as 2 (M) x 4 (N), each owns a 128 x 64 sub-tile (4 x 2 accumulators of 32 x 32 = 128
// registers), and every wave both stages and computes.  Per K-tile a wave requests its 8 pieces (4 of A, 4 of W,
// 1 KB = 8 rows x 128 B each) with MUBUF buffer_load_dwordx4 (SGPR descriptor + 32-bit lane offset + scalar
// K offset: no VALU per load, and unlike FLAT loads they issue next to a busy matrix pipe) into ONE 32-register
// set one K-tile ahead, and writes them with ds_write_b128 into the other half of a 2-stage LDS ring during the
// current K-tile: the A pieces in its first half, the W pieces in its second, each followed at once by the
// requests for the K-tile after (hand-counted vmcnt: loads return in order).  One s_barrier per K-tile; fragment
// reads are software-pipelined over the four 16-deep sub-steps (two register sets).  LDS image, swizzle and MFMA operand order are those of the 128 x 128 kernel, so every output is
// BIT-IDENTICAL to it (same products, same accumulation order): tests/test_gpu_gemm256.py compares the two.
//
// Epilogue: wave-private.  Each wave pushes its sub-tile through its own 16 KB slice of the (now idle) stage
// ring, 32 rows at a time, so that every global store is a 16-byte lane-contiguous piece of a 128-byte (f16) or
// 256-byte (f32) row segment; no block barrier after the K loop's last one, waves drift apart and one wave's
// stores overlap another's conversion VALU.

#include "qv_kernels.h"
#include "qv_gemm_dequant.h"
#include "qv_ort.h"
#include "qv_dev_util.h"

#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef half2_t h2_t;

__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

constexpr bool out_is_f32(int epi) { return epi == EPI_RESID || epi == EPI_F32; }

#ifdef QV_GEMM_TRACE   // dev tool only (tools/gemm256_trace.hip)
#define QW_ABL(bit) (g.abl & (bit))
#define QW_PHASE(slot) do { if (g.phase && tid == 0) g.phase[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (slot)] = wall_clock64(); } while (0)
#define QW_TRACE(kt) do { if (g.trace && lane == 0) g.trace[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 64 + (kt)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define QW_ABL(bit) false
#define QW_PHASE(slot) do { } while (0)
#define QW_TRACE(kt) do { } while (0)
#endif

}  // namespace

// WQ: 0 = f16 weights, 4 = W4A16 (block-128 int4), 8 = W8A16 (per-channel int8), 88 = A8W8 (QV_PREC_ORT_MIXED: s8
// activations x s8 weights on v_mfma_i32_32x32x32_i8 -- K counts byte pairs, so staging, LDS image and fragment reads are
// the f16 kernel's unchanged; only the accumulator type and the epilogue differ); layouts in qv_kernels.h (GemmArgs)
// MI: 32-row accumulator fragments per wave = tile height / 64: 4 = the 256 x 256 tile, 3 = a 192 x 256 tile (round 6).  The
// shorter tile exists for the row counts whose 256-row grid leaves a quarter of the chip idle -- M = 24,064 (64 clips of
// 30 s): 94 x 2 = 188 tiles of 256 rows on 256 CUs for every N = 512 GEMM, 126 x 2 = 252 tiles of 192 rows -- at 17 % more
// operand bytes per flop.  Products and accumulation order per output element do not depend on MI: the bits are the same.
template <int EPI, int WQ, int MI>
__global__ __launch_bounds__(512, 1) void k_gemm256(GemmArgs g) {
    constexpr bool W4 = WQ == 4, W8 = WQ == 8, I8 = WQ == 88;
    constexpr int BM = 64 * MI, BN = 256, BK = 64;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = W4 ? BN * BK / 2 : W8 ? BN * BK : BN * BK * 2;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NBP = W4 ? 1 : W8 ? 2 : 4;   // 1 KB pieces of the W tile per wave per K-tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    QW_PHASE(0);
    // XCD-aware tile order (bijective for any tile count): consecutive workgroup ids land on different XCDs;
    // each XCD gets a contiguous run of tiles, which share A row panels in its private L2
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
    acc_t acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    // ------------------------------------------------------------------ staging ----------
    // piece q of this wave: tile rows wave*32 + q*8 .. +7 of A (and of W), 128 B per row; lane -> (row lane>>3,
    // 16-byte chunk lane&7).  LDS rows are 128 B, chunk c of row r sits at c ^ ((r >> 1) & 7) (conflict-free
    // ds_read_b128 fragment reads, same image as qv_gemm.hip).
    const int nk = g.K / BK;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)g.A, 0, (int)((size_t)g.M * g.lda * 2), 0x00020000);
    const void *bbase = W4 ? (const void *)g.Wq : W8 ? (const void *)g.W8 : I8 ? (const void *)g.Wi8 : (const void *)g.W;
    const size_t bbytes = W4 ? (size_t)g.N * g.K / 2 : W8 ? (size_t)g.N * g.K : (size_t)g.N * g.ldw * 2;
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)bbase, 0, (int)bbytes, 0x00020000);
    constexpr int stepB = W4 ? 2048 : W8 ? 4096 : BK * 2;   // bytes between consecutive K-tiles of a W piece
    unsigned offA[MI], offB[NBP];
    int dst[MI], dstB[NBP];
#pragma unroll
    for (int q = 0; q < MI; ++q) {
        const int row = wave * (8 * MI) + q * 8 + (lane >> 3), c = lane & 7;
        int grow = (QW_ABL(64) ? 0 : m0) + row;   // (ablation 64: every block loads tile (0, 0) -- all loads hit L2)
        grow = grow < g.M ? grow : g.M - 1;   // rows past M repeat the last one; their outputs are never stored
        offA[q] = (unsigned)(((size_t)grow * g.lda + c * 8) * 2);
        dst[q] = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
    }
    if (!W4 && !W8) {
#pragma unroll
        for (int q = 0; q < (W4 || W8 ? 0 : 4); ++q) {
            const int row = wave * 32 + q * 8 + (lane >> 3), c = lane & 7;
            offB[q] = (unsigned)(((size_t)((QW_ABL(64) ? 0 : n0) + row) * g.ldw + c * 8) * 2);
            dstB[q] = A_BYTES + row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
        }
    }
    if (W4) {
        // 64 x 64 nibble tiles, 2 KB contiguous and already in the LDS order: one 1 KB piece (32 tile rows) per wave
        offB[0] = (unsigned)(((size_t)((n0 >> 6) + (wave >> 1)) * nk) * 2048 + (wave & 1) * 1024 + lane * 16);
        dstB[0] = A_BYTES + wave * 1024 + lane * 16;
    }
    if (W8) {
        // 64 x 64 byte tiles, 4 KB contiguous and already in the LDS order: two 1 KB pieces (16 tile rows each) per wave
#pragma unroll
        for (int q = 0; q < (W8 ? 2 : 0); ++q) {
            const int pc = wave * 2 + q;
            offB[q] = (unsigned)(((size_t)((n0 >> 6) + (pc >> 2)) * nk) * 4096 + (pc & 3) * 1024 + lane * 16);
            dstB[q] = A_BYTES + pc * 1024 + lane * 16;
        }
    }
    u32x4 ra[MI], rb[NBP];
    auto fetchA = [&](int kt) {
        if (QW_ABL(8)) return;
        const int so = kt * (BK * 2);
#pragma unroll
        for (int q = 0; q < MI; ++q)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ra[q]) : "v"(offA[q]), "s"(rsA), "s"(so) : "memory");
    };
    auto fetchB = [&](int kt) {
        if (QW_ABL(8)) return;
        const int so = kt * stepB;
#pragma unroll
        for (int q = 0; q < NBP; ++q)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rb[q]) : "v"(offB[q]), "s"(rsB), "s"(so) : "memory");
    };
    auto putA = [&](int stage) {
        if (QW_ABL(4)) return;
        unsigned char *st = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < MI; ++q) *(u32x4 *)(st + dst[q]) = ra[q];
    };
    auto putB = [&](int stage) {
        if (QW_ABL(4)) return;
        unsigned char *st = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < NBP; ++q) *(u32x4 *)(st + dstB[q]) = rb[q];
    };

    // write piece q, re-request it for the tile after: alternating the two keeps the LDS store path and the
    // address path busy at the same time (4 stores then 4 loads queue up behind one, then the other)
    auto swapA = [&](int stage, int kt) {
        unsigned char *st = smem + stage * STAGE_BYTES;
        const int so = kt * (BK * 2);
#pragma unroll
        for (int q = 0; q < MI; ++q) {
            if (!QW_ABL(4)) *(u32x4 *)(st + dst[q]) = ra[q];
            if (!QW_ABL(8)) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ra[q]) : "v"(offA[q]), "s"(rsA), "s"(so) : "memory");
        }
    };
    auto swapB = [&](int stage, int kt) {
        unsigned char *st = smem + stage * STAGE_BYTES;
        const int so = kt * stepB;
#pragma unroll
        for (int q = 0; q < NBP; ++q) {
            if (!QW_ABL(4)) *(u32x4 *)(st + dstB[q]) = rb[q];
            if (!QW_ABL(8)) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rb[q]) : "v"(offB[q]), "s"(rsB), "s"(so) : "memory");
        }
    };

    // W4: this tile's {scale, 1024 + zero point} pairs [K/128][BN] stay in LDS behind the stage ring for the whole K loop
    h2_t *sS = (h2_t *)(smem + 2 * STAGE_BYTES);
    if (W4) {
        const int nkb = g.K >> 7;
        for (int idx = tid; idx < BN * nkb; idx += 512) {
            const int kb = idx / BN, n = idx - kb * BN;
            sS[idx] = ((const h2_t *)g.wscale)[(size_t)kb * g.N + n0 + n];
        }
    }

    fetchA(0);
    fetchB(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    putA(0);
    putB(0);
    if (nk > 1) {
        fetchA(1);
        fetchB(1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    struct Frag { half8 a[MI]; half8 b[2]; uint32_t q4[2]; uint2 q8[2]; };
    QW_PHASE(1);
    // All eight waves run the same phase.  Fragment reads are software-pipelined under the wave's own MFMAs (two
    // register sets, pinned with sched_barrier); the A pieces of tile kt + 1 are written (and the A pieces of tile
    // kt + 2 requested) behind the first 8 MFMAs of K-tile kt, the W pieces behind the second 8; one barrier per
    // K-tile.  Measured against this (tools/lds_mfma_bench.hip, DESIGN.md "GEMM, round 2b"): the two halves of the
    // block running one phase apart (ping-pong: memory phase of waves 0..3 beside the MFMA phase of waves 4..7) and
    // the 8 loads spread two per MFMA group -- neither was faster.
    auto rd = [&](int stage, int ks, Frag &f) {
        if (QW_ABL(2)) return;
        const half_t *sA = (const half_t *)(smem + stage * STAGE_BYTES);
        const half_t *sB = (const half_t *)((const unsigned char *)sA + A_BYTES);
        const int c = ks * 2 + (lane >> 5);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wn * 64 + j * 32 + (lane & 31);
            if (W4)        // 32-byte rows, 4-byte chunk c at c ^ ((row >> 2) & 7): conflict-free ds_read_b32
                f.q4[j] = *(const uint32_t *)((const unsigned char *)sB + row * 32 + ((c ^ ((row >> 2) & 7)) << 2));
            else if (W8)   // 64-byte rows in 4 KB tiles of 64 rows, 8-byte chunk c at c ^ ((row >> 2) & 7); bytes are q + 128
                f.q8[j] = *(const uint2 *)((const unsigned char *)sB + (row >> 6) * 4096 + (row & 63) * 64 + ((c ^ ((row >> 2) & 7)) << 3));
            else
                f.b[j] = *(const half8 *)(sB + row * 64 + ((c ^ ((row >> 1) & 7)) << 3));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = wm * (32 * MI) + i * 32 + (lane & 31);
            f.a[i] = *(const half8 *)(sA + row * 64 + ((c ^ ((row >> 1) & 7)) << 3));
        }
    };
    h2_t sc[2], zo[2];   // W4: this K-tile's scale / offset pairs of the wave's two column fragments
    auto mma = [&](const Frag &f) {
        half8 b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = W4 ? dequant8(f.q4[j], sc[j], zo[j]) : W8 ? dequant8_i8(f.q8[j]) : f.b[j];
#if defined(QV_GEMM_TRACE) && defined(__HIP_DEVICE_COMPILE__)
        if (QW_ABL(1)) {
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(f.a[i]));
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(b[j]));
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // operands swapped (D^T = W A^T): a lane holds 4 CONSECUTIVE output columns per register quad
                if constexpr (I8)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, b[j]), __builtin_bit_cast(i32x4, f.a[i]), acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], f.a[i], acc[i][j], 0, 0, 0);
            }
    };
    // accumulator (i, j), register r: tile row = wm*32*MI + i*32 + (lane & 31),
    //   tile column = wn*64 + j*32 + 8*(r >> 2) + 4*(lane >> 5) + (r & 3)
    const int l31 = lane & 31, hi = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc((void *)g.bias, 0, g.N * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_scl = __builtin_amdgcn_make_buffer_rsrc((void *)(W8 ? g.w8scale : g.bias), 0, g.N * 4, 0x00020000);
    auto ldf4 = [&](const __amdgpu_buffer_rsrc_t &rs, int elem) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, elem * 4, 0, 0));
    };
    const int nb = n0 + wn * 64;          // first tile column of this wave
    const int mb = m0 + wm * (32 * MI);   // first row of this wave
    f32x4 bia[2][4], scl[2][4];           // bias (and, W8A16, per-channel weight scales) of the wave's 64 columns

    // The last K-tile is written out separately (it stages nothing): +3 % on the long-K residual shapes.  Requesting
    // the epilogue's bias vectors at its top, and K-tiles 0 and 1 back to back at start-up, were measured on top of
    // that and are not in: neutral at M = 8064, -4 % on FFN-up / GLU at M = 32,256.
    auto ktile = [&](int kt, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        QW_TRACE(kt);
        const int cur = kt & 1;
        const bool has2 = kt + 2 < nk;
        Frag f0 = {}, f1 = {};
        if (W4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const h2_t sz = sS[(kt >> 1) * BN + wn * 64 + j * 32 + (lane & 31)];
                sc[j] = h2_t{sz[0], sz[0]};
                zo[j] = h2_t{sz[1], sz[1]};
            }
        }
        rd(cur, 0, f0);
        rd(cur, 1, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(f0);
        __builtin_amdgcn_sched_barrier(0);
        if (!LAST) {
            // outstanding, oldest first: A(kt+1) x4, W(kt+1) x NBP
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP) : "memory");
            if (has2 && QV_SWAP) swapA(cur ^ 1, kt + 2);
            else { putA(cur ^ 1); if (has2) fetchA(kt + 2); }
        }
        __builtin_amdgcn_sched_barrier(0);
        rd(cur, 2, f0);
        __builtin_amdgcn_sched_barrier(0);
        mma(f1);
        __builtin_amdgcn_sched_barrier(0);
        if (!LAST) {
            // outstanding: W(kt+1) x NBP [, A(kt+2) x MI]
            if (has2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (has2 && QV_SWAP) swapB(cur ^ 1, kt + 2);
            else { putB(cur ^ 1); if (has2) fetchB(kt + 2); }
        }
        __builtin_amdgcn_sched_barrier(0);
        rd(cur, 3, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(f0);
        __builtin_amdgcn_sched_barrier(0);
        mma(f1);
        // stage kt + 1 written (own ds_writes retired) and stage kt read by every wave
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    for (int kt = 0; kt + 1 < nk; ++kt) ktile(kt, std::false_type{});
    ktile(nk - 1, std::true_type{});
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bia[j][q] = ldf4(rs_bias, nb + j * 32 + 8 * q + 4 * hi);
            if (W8) scl[j][q] = ldf4(rs_scl, nb + j * 32 + 8 * q + 4 * hi);
        }
    QW_TRACE(nk);
    QW_PHASE(2);
#if defined(QV_GEMM_TRACE) && defined(__HIP_DEVICE_COMPILE__)
    if (QW_ABL(32)) {   // no epilogue at all (accumulators kept live)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
#endif
    // ------------------------------------------------------------------ epilogue ----------
    unsigned char *sW = smem + wave * 16384;   // this wave's private staging slice

    if constexpr (I8) {
        // ---- A8W8 epilogue (qv_ort.h; same arithmetic as k_gemm<.., 88, ..>, bit for bit):
        //   v = float(acc + (128 - zp_x[u]) * wsum[n]) * (s_x[u] * s_w) + bias[n],  u = the row's utterance
        // wave-private like the f16 epilogues: 32 rows at a time through the wave's own LDS slice.
        constexpr bool GLU = EPI == EPI_GLU, OUT16 = EPI == EPI_F16_RELU;
        constexpr bool RELU = EPI == EPI_F16_RELU || EPI == EPI_F32_RELU;
        constexpr bool FOLD = EPI == EPI_GLU || EPI == EPI_F32_RELU;   // the output feeds another quantiser: track its range
        constexpr int WO = GLU ? 32 : 64;                               // output columns of this wave
        constexpr int LDT = WO + 4;
        float *sO = (float *)sW;
        const __amdgpu_buffer_rsrc_t rs_ws = __builtin_amdgcn_make_buffer_rsrc((void *)g.wsum, 0, g.N * 4, 0x00020000);
        i32x4 wsm[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                wsm[j][q] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ws, (nb + j * 32 + 8 * q + 4 * hi) * 4, 0, 0));
        auto row_owner = [&](int grow, bool &valid) {
            valid = true;
            if (g.row_map) return g.row_map[grow] >> 16;
            const int u = grow / g.rows_per_utt;
            valid = (grow - u * g.rows_per_utt) / g.f_per_t < g.len[u];
            return u;
        };
        const int nbo = GLU ? n0 / 2 + wn * 32 : nb;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            // the tile's rows are rebased per 32-row group: the dense subsampling tensors have millions of 1 KB rows
            const int r0 = mb + i * 32;
            const int rows_here = g.M - r0 < 32 ? g.M - r0 : 32;
            if (rows_here <= 0) break;
            int grow = r0 + l31;
            grow = grow < g.M ? grow : g.M - 1;
            bool valid;
            const int utt = row_owner(grow, valid);
            const QParam p = dql_param(g.mm_in + QV_MM_STRIDE * utt);
            const float srow = p.scale * g.w_scale;
            const int zc = 128 - (int)p.zp;
            float mn = INFINITY, mx = -INFINITY;
            f32x4 old[8];
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
                OUT16 ? (void *)((half_t *)g.out + (size_t)r0 * g.ldo) : (void *)((float *)g.out + (size_t)r0 * g.ldo), 0,
                rows_here * g.ldo * (OUT16 ? 2 : 4), 0x00020000);
            if (EPI == EPI_RESID) {
                const int rr = lane >> 4, cc = (lane & 15) * 4;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = k * 4 + rr;
                    if (r < rows_here) old[k] = ldf4(rs_out, r * g.ldo + nbo + cc);
                }
            }
            if (GLU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 av, gv, o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        av[e] = (float)(acc[i][0][q * 4 + e] + zc * wsm[0][q][e]) * srow + bia[0][q][e];
                        gv[e] = (float)(acc[i][1][q * 4 + e] + zc * wsm[1][q][e]) * srow + bia[1][q][e];
                    }
                    const f32x4 sg = sigmoid4(gv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = av[e] * sg[e]; mn = fminf(mn, o[e]); mx = fmaxf(mx, o[e]); }
                    *(f32x4 *)(sO + l31 * LDT + 8 * q + 4 * hi) = o;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = (float)(acc[i][j][q * 4 + e] + zc * wsm[j][q][e]) * srow + bia[j][q][e];
                            if (RELU) v = v > 0.f ? v : 0.f;
                            o[e] = v;
                            mn = fminf(mn, v);
                            mx = fmaxf(mx, v);
                        }
                        *(f32x4 *)(sO + l31 * LDT + j * 32 + 8 * q + 4 * hi) = o;
                    }
            }
            if (FOLD) {
                // range of each row (both lane halves), then one atomic pair per run of rows of the same utterance
                mn = fminf(mn, __shfl_xor(mn, 32));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const int u = (valid && r0 + l31 < g.M) ? utt : -1;
                unsigned long long todo = __ballot(hi == 0 && u >= 0);
                while (todo) {   // (wave-uniform: one round per distinct utterance among the 32 rows, usually one)
                    const int first = __ffsll((long long)todo) - 1;
                    const int uu = __shfl(u, first);
                    const bool in = hi == 0 && u == uu;
                    const float a = wave_min(in ? mn : INFINITY), b2 = wave_max(in ? mx : -INFINITY);
                    if (lane == first) mm_fold(g.mm_out + QV_MM_STRIDE * uu, a, b2);
                    todo &= ~__ballot(in);
                }
            }
            if (OUT16) {
                const int rr = lane >> 3, cc = (lane & 7) * 8;    // read-back: 8 rows x 128 B of halves per instruction
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = k * 8 + rr;
                    const f32x4 a = *(const f32x4 *)(sO + r * LDT + cc), b2 = *(const f32x4 *)(sO + r * LDT + cc + 4);
                    const half8 h = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b2[0], (half_t)b2[1], (half_t)b2[2], (half_t)b2[3]};
                    if (r < rows_here) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs_out, (r * g.ldo + nbo + cc) * 2, 0, 0);
                }
            } else if (GLU) {
                const int rr = lane >> 3, cc = (lane & 7) * 4;    // read-back: 8 rows x 128 B per instruction
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = k * 8 + rr;
                    const f32x4 v = *(const f32x4 *)(sO + r * LDT + cc);
                    if (r < rows_here) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_out, (r * g.ldo + nbo + cc) * 4, 0, 0);
                }
            } else {
                const int rr = lane >> 4, cc = (lane & 15) * 4;   // read-back: 4 rows x 256 B per instruction
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = k * 4 + rr;
                    f32x4 v = *(const f32x4 *)(sO + r * LDT + cc);
                    if (EPI == EPI_RESID) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = old[k][e] + g.alpha * v[e];
                    }
                    if (r < rows_here) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_out, (r * g.ldo + nbo + cc) * 4, 0, 0);
                }
            }
        }
        return;
    }

    if (EPI == EPI_QKV && n0 >= 2 * QV_D) {
        // V tile: stored TRANSPOSED, Vt[b][h*64+d][t].  64 frames at a time go through the wave's slice as
        // [d][frame]; a store instruction then covers two d rows x 64 consecutive frames (4-byte frame pairs).
        constexpr int LDV = 64 + 2;   // halves per d row (odd dword pitch)
        half_t *sT = (half_t *)sW;
        half_t *vt = (half_t *)g.out2;
        const int fp = lane & 31, dsub = lane >> 5;
#pragma unroll
        for (int hh = 0; hh < (MI + 1) / 2; ++hh) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
                if (hh * 2 + ii < MI) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                sT[(j * 32 + 8 * q + 4 * hi + e) * LDV + ii * 32 + l31] = (half_t)(acc[hh * 2 + ii][j][q * 4 + e] + bia[j][q][e]);
                }
            // (MI = 3: the second 64-frame group holds one 32-row fragment -- frames 32 .. 63 of it belong to the next wave)
            const bool mine = hh * 64 + 2 * fp + 1 < 32 * MI;
            const int r0 = mb + hh * 64 + 2 * fp, r1 = r0 + 1;
            const int bt0 = (mine && r0 < g.M) ? g.row_map[r0] : -1, bt1 = (mine && r1 < g.M) ? g.row_map[r1] : -1;
            const bool pair = bt0 >= 0 && bt1 == bt0 + 1 && (bt0 & 1) == 0;   // same utterance, even frame: one 4-byte store
            for (int dd = 0; dd < 32; ++dd) {
                const int d = dd * 2 + dsub;
                const size_t drow = (size_t)(nb - 2 * QV_D + d);
                const half_t v0 = sT[d * LDV + 2 * fp], v1 = sT[d * LDV + 2 * fp + 1];
                if (pair) {
                    h2_t v = {v0, v1};
                    *(h2_t *)(vt + ((size_t)(bt0 >> 16) * QV_D + drow) * g.t_pad + (bt0 & 0xFFFF)) = v;
                } else {
                    if (bt0 >= 0) vt[((size_t)(bt0 >> 16) * QV_D + drow) * g.t_pad + (bt0 & 0xFFFF)] = v0;
                    if (bt1 >= 0) vt[((size_t)(bt1 >> 16) * QV_D + drow) * g.t_pad + (bt1 & 0xFFFF)] = v1;
                }
            }
        }
        return;
    }

    if (out_is_f32(EPI)) {
        constexpr int LDT = 64 + 4;   // floats per staged row
        float *sO = (float *)sW;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)((size_t)g.M * g.ldo * 4), 0x00020000);
        const int rr = lane >> 4, cc = (lane & 15) * 4;   // read-back: 4 rows x 256 B per instruction
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            // (requesting row group i + 1's old values before group i is stored -- two register sets -- was measured:
            // no gain, the other waves' epilogues already cover the read latency)
            f32x4 old[8];
            if (EPI == EPI_RESID) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = mb + i * 32 + k * 4 + rr;
                    if (r < g.M && !QW_ABL(16)) old[k] = ldf4(rs_out, r * g.ldo + nb + cc);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[i][j][q * 4 + e];
                        if (W8) x *= scl[j][q][e];
                        v[e] = g.alpha * (x + bia[j][q][e]);
                    }
                    *(f32x4 *)(sO + l31 * LDT + j * 32 + 8 * q + 4 * hi) = v;
                }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = mb + i * 32 + k * 4 + rr;
                f32x4 v = *(const f32x4 *)(sO + (k * 4 + rr) * LDT + cc);
                if (r >= g.M || QW_ABL(16)) { asm volatile("" ::"v"(v)); continue; }
                if (EPI == EPI_RESID) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += old[k][e];
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_out, (r * g.ldo + nb + cc) * 4, 0, 0);
            }
        }
        return;
    }

    if (EPI == EPI_GLU) {
        // W rows interleaved in 32-channel groups, [value(32) | gate(32)] per 64 columns: this wave's two
        // accumulator columns are one value / gate pair -> 32 output channels
        constexpr int LDT = 32 + 8;
        half_t *sO = (half_t *)sW;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)((size_t)g.M * g.ldo * 2), 0x00020000);
        const int rr = lane >> 2, cc = (lane & 3) * 8;    // read-back: 16 rows x 64 B per instruction
        const int nbo = n0 / 2 + wn * 32;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                half4 o;
                f32x4 av, gv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    av[e] = acc[i][0][q * 4 + e];
                    gv[e] = acc[i][1][q * 4 + e];
                    if (W8) { av[e] *= scl[0][q][e]; gv[e] *= scl[1][q][e]; }
                    av[e] += bia[0][q][e];
                    gv[e] += bia[1][q][e];
                }
                const f32x4 sg = sigmoid4(gv);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)(av[e] * sg[e]);
                *(half4 *)(sO + l31 * LDT + 8 * q + 4 * hi) = o;
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = mb + i * 32 + k * 16 + rr;
                const u32x4 v = *(const u32x4 *)(sO + (k * 16 + rr) * LDT + cc);
                if (r < g.M && !QW_ABL(16)) __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, (r * g.ldo + nbo + cc) * 2, 0, 0);
            }
        }
        return;
    }

    {
        constexpr int LDT = 64 + 8;   // halves per staged row
        half_t *sO = (half_t *)sW;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)((size_t)g.M * g.ldo * 2), 0x00020000);
        const int rr = lane >> 3, cc = (lane & 7) * 8;    // read-back: 8 rows x 128 B per instruction
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    half4 o;
                    f32x4 xv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[e] = acc[i][j][q * 4 + e] + bia[j][q][e];
                    if (EPI == EPI_F16_SWISH) xv = swish4(xv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = xv[e];
                        if (EPI == EPI_F16_RELU) x = x > 0.f ? x : 0.f;
                        o[e] = (half_t)x;
                    }
                    *(half4 *)(sO + l31 * LDT + j * 32 + 8 * q + 4 * hi) = o;
                }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = mb + i * 32 + k * 8 + rr;
                const u32x4 v = *(const u32x4 *)(sO + (k * 8 + rr) * LDT + cc);
                if (r < g.M && !QW_ABL(16)) __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, (r * g.ldo + nb + cc) * 2, 0, 0);
            }
        }
    }
}

// (tools/gemm256q.h: a four-wave variant of this kernel, measured slower in round 5; only tools/gemm_bench compiles it)
#ifdef QV_GEMM_Q_VARIANT
#include QV_GEMM_Q_VARIANT
#endif

template <int EPI, int WQ, int MI>
static void launch256m(const GemmArgs &g, hipStream_t s) {
    constexpr int LDS = 2 * (256 * 64 * 2) * 2;   // two stages of f16 A + W tiles = 128 KB = the eight epilogue slices
#ifdef QV_GEMM_Q_VARIANT
    if constexpr (WQ != 88 && MI == 4) {
        if (gemm_q()) {
            static bool opted_q = false;
            if (!opted_q) {
                (void)hipFuncSetAttribute((const void *)k_gemm256q<EPI, WQ>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                opted_q = true;
            }
            hipLaunchKernelGGL((k_gemm256q<EPI, WQ>), dim3(g.N / 256, (g.M + 255) / 256), dim3(256), LDS, s, g);
            return;
        }
    }
#endif
    static bool opted = false;
    if (!opted) {
        (void)hipFuncSetAttribute((const void *)k_gemm256<EPI, WQ, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        opted = true;
    }
    hipLaunchKernelGGL((k_gemm256<EPI, WQ, MI>), dim3(g.N / 256, (g.M + 64 * MI - 1) / (64 * MI)), dim3(512), LDS, s, g);
}

static thread_local int t_bm = 256;   // tile height of the launch in progress (launch_gemm256's argument)
template <int EPI, int WQ>
static void launch256(const GemmArgs &g, hipStream_t s) {
    if (t_bm == 192) launch256m<EPI, WQ, 3>(g, s);
    else launch256m<EPI, WQ, 4>(g, s);
}

// N % 256 == 0, K % 64 == 0 (int4: K % 128 == 0, K <= 4096 for the scale table in LDS); false = nothing launched
bool launch_gemm256(int epi, const GemmArgs &g, hipStream_t s, int bm) {
    if (g.N % 256 != 0 || g.K % 64 != 0 || !g.bias || (bm != 256 && bm != 192)) return false;
    t_bm = bm;
    if (g.Wi8) {
        // int8 activations x int8 weights (QV_PREC_ORT_MIXED): the GEMM-shaped convolutions
        switch (epi) {
            case EPI_GLU: launch256<EPI_GLU, 88>(g, s); break;
            case EPI_RESID: launch256<EPI_RESID, 88>(g, s); break;
            case EPI_F32: launch256<EPI_F32, 88>(g, s); break;
            case EPI_F32_RELU: launch256<EPI_F32_RELU, 88>(g, s); break;
            case EPI_F16_RELU: launch256<EPI_F16_RELU, 88>(g, s); break;
            default: return false;
        }
        return true;
    }
    if (g.Wq) {
        // int4 weights: the Linear layers (FFN, QKV, attention out, linear_pos)
        if (g.K % 128 != 0 || g.K > 4096) return false;
        switch (epi) {
            case EPI_F16: launch256<EPI_F16, 4>(g, s); break;
            case EPI_F16_SWISH: launch256<EPI_F16_SWISH, 4>(g, s); break;
            case EPI_RESID: launch256<EPI_RESID, 4>(g, s); break;
            case EPI_QKV: launch256<EPI_QKV, 4>(g, s); break;
            default: return false;
        }
        return true;
    }
    if (g.W8) {
        // int8 weights: the pointwise convolutions of the conv module
        switch (epi) {
            case EPI_GLU: launch256<EPI_GLU, 8>(g, s); break;
            case EPI_RESID: launch256<EPI_RESID, 8>(g, s); break;
            default: return false;
        }
        return true;
    }
    switch (epi) {
        case EPI_F16: launch256<EPI_F16, 0>(g, s); break;
        case EPI_F16_SWISH: launch256<EPI_F16_SWISH, 0>(g, s); break;
        case EPI_F16_RELU: launch256<EPI_F16_RELU, 0>(g, s); break;
        case EPI_RESID: launch256<EPI_RESID, 0>(g, s); break;
        case EPI_F32: launch256<EPI_F32, 0>(g, s); break;
        case EPI_QKV: launch256<EPI_QKV, 0>(g, s); break;
        case EPI_GLU: launch256<EPI_GLU, 0>(g, s); break;
        default: return false;
    }
    return true;
}
