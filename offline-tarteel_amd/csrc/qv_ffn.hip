// qv_ffn.hip -- fused Conformer feed-forward module (prototype of VERDICT r3 item 3, measured by tools/ffn_fused_bench.hip):
//
//     out[M,512] += alpha * ( swish(X[M,512] * W1^T + b1)[M,2048] * W2^T + b2 )
//
// in ONE kernel: the hidden activation goes from the first GEMM's accumulators straight into the second GEMM's operand
// registers -- it touches neither HBM nor LDS.
//
// Why "token-stationary".  The module's working set per 128 tokens is X (128 KB f16) + the output accumulators (256 KB
// f32): more than the LDS (160 KB), exactly the register file (512 KB) minus what the loop needs.  Splitting the OUTPUT
// columns over waves (the 256 x 256 GEMM's 2 x 4 wave grid) makes every wave need the whole hidden chunk of its rows, i.e.
// an exchange through LDS plus X re-streamed per chunk (47 B/clk of global->LDS traffic at full MFMA rate against the
// 64 B/clk the CU's address path moves: worse than the two-kernel path).  So here a wave owns 32 TOKENS and everything
// that belongs to them:
//   * its X rows as 32 MFMA operand fragments in registers (128 VGPRs), loaded once;
//   * its 32 x 512 output tile as 16 accumulators (256 AGPRs) for the whole kernel;
//   * per 32 hidden channels: GEMM1 = 32 MFMAs into one 32 x 32 accumulator, bias + Swish + f16 in registers, and the
//     result IS the operand of GEMM2's 32 MFMAs (the contraction index of GEMM2 is permuted to the accumulator layout;
//     W2 is packed with the same permutation on the host, csrc/qv_ffn.h).
// Four waves (one per SIMD, <= 512 registers each) = 128 tokens per block; the only shared resource is the weight
// stream: 256 units of 16 KB (= 16 MFMA fragments, stored fragment-major so that the loader is a linear copy and a
// fragment read is one conflict-free ds_read_b128 at an immediate offset), staged global -> registers -> LDS two units
// ahead by all four waves, one s_barrier per unit (16 MFMAs per wave).  Per MAC this moves the same weight bytes as the
// 256 x 256 GEMM tile moves operand bytes (32 B/clk at full rate), reads 1 KB of LDS per MFMA (128 B/clk of 256) and
// writes nothing but the weight ring -- and pays one prologue / epilogue per 537 MFLOP instead of per 67.
//
// Grid = ceil(M / 128) blocks: 63 at B = 64 x 10 s (a quarter of the chip: meant to run beside other batches' kernels,
// like the 64-tile FFN-down it replaces), 252 at B = 256.

#include "qv_ffn.h"
#include "qv_dev_util.h"

#include <stdio.h>

#include <type_traits>
#include <vector>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NSLOT = 4;                      // LDS ring slots of one unit each
constexpr int UNIT = QV_FFN_UNIT_BYTES;
constexpr int NCHUNK = QV_FF / 32;            // 64 chunks of 32 hidden channels

__device__ __forceinline__ float ffn_sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// scheduling-group masks of __builtin_amdgcn_sched_group_barrier
#define SG_VALU 0x002
#define SG_MFMA 0x008
#define SG_DSR 0x100
#ifdef QV_FFN_NOSG
#define QV_SGB(mask, n) do { } while (0)
#else
#define QV_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#endif
#ifdef QV_FFN_NOSB
#define QV_SB() do { } while (0)
#else
#define QV_SB() __builtin_amdgcn_sched_barrier(0)
#endif

// ABL: timing ablations for tools/ffn_fused_bench.hip (results are WRONG with any bit set; the product launches ABL = 0):
//   1 no activation VALU, 2 GEMM1 without the accumulate dependency (C = 0 every MFMA), 4 no s_barrier, 8 no fragment reads,
//   16 no weight staging (no global loads, no ds_writes), 32 GEMM2 without the accumulate dependency
template <int ABL>
__global__ __launch_bounds__(256) void k_ffn_fused(FfnArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *sB1 = (float *)(smem + NSLOT * UNIT);   // b1 [2048] behind the ring
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tok = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.x * 128 + wave * 32;
    int row = m0 + tok;
    row = row < g.M ? row : g.M - 1;          // rows past M repeat the last one; their outputs are never stored

    // ---- weight stream: unit u = 16 KB at Wp + u * 16 KB; this wave copies 4 KB of it (4 pieces of 1 KB).  Units past
    // the end read as zeros (buffer bounds check) and land in a ring slot nobody reads: no branch in the steady state.
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)g.Wp, 0, QV_FFN_UNITS * UNIT, 0x00020000);
    const unsigned lofs = (unsigned)(wave * 4096 + lane * 16);
    u32x4 st[4];
    auto fetch = [&](int u) {
        if (ABL & 16) { if (u > 2) return; }
#pragma unroll
        for (int i = 0; i < 4; ++i) st[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, lofs + i * 1024, u * UNIT, 0);
    };
    auto put = [&](int u) {
        if (ABL & 16) { if (u > 2) return; }
        unsigned char *s = smem + (u & (NSLOT - 1)) * UNIT + lofs;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x4 *)(s + i * 1024) = st[i];
    };
    fetch(0);
    for (int i = tid; i < QV_FF / 4; i += 256) ((f32x4 *)sB1)[i] = ((const f32x4 *)g.b1)[i];

    // ---- this wave's X rows as MFMA fragments: fragment ks = k 16 ks .. 16 ks + 15, lane (token, hi) holds 8 hi .. 8 hi + 7
    half8 xf[32];
    {
        const half_t *xp = g.X + (size_t)row * g.ldx + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) xf[ks] = *(const half8 *)(xp + ks * 16);
    }
    f32x16 acc2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;

    put(0);
    fetch(1);
    put(1);
    fetch(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // One step = one unit = 16 MFMAs.  Units u and u + 1 are published when step u starts; the step writes unit u + 2 (in
    // the staging registers since the step before) into the ring, requests unit u + 3, and reads fragments 4..15 of unit u
    // plus 0..3 of unit u + 1 four at a time, each group one MFMA group ahead of its use: the LDS latency of a unit's first
    // fragments is paid under the previous unit's MFMAs, not behind the barrier.
    int u = 0;
    half8 fr[4];
    auto rd1 = [&](half8 &f, int unit, int fi) {
        if (ABL & 8) { if (unit > 0) return; }
        f = *(const half8 *)(smem + (unit & (NSLOT - 1)) * UNIT + lane * 16 + fi * 1024);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) rd1(fr[i], 0, i);
    auto step_end = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
        ++u;
    };
    const std::integral_constant<int, 0> C0{};
    const std::integral_constant<int, 1> C1{};

    // bias + Swish + f16 of two accumulator registers pairs (q = 2 part + {0, 1}, element e): they are k-slots e and 4 + e
    // of k-step `part` of GEMM2's operand (lane (token, hi), register 4 q + e <-> hidden channel 8 q + 4 hi + e of the chunk)
    auto act1 = [&](const f32x16 &acc, const f32x4 &ba, const f32x4 &bb, int part, int e, half8 &out) {
        const float a = acc[8 * part + e] + ba[e], b = acc[8 * part + 4 + e] + bb[e];
        out[e] = (half_t)(a * ffn_sigm(a));
        out[4 + e] = (half_t)(b * ffn_sigm(b));
    };

    // GEMM1 step: K half kh of the NEXT chunk's hidden tile into `nxt`; meanwhile (ACT) half of the CURRENT chunk's
    // accumulator `cur` is activated into hf[kh] -- its VALU work is spread between the MFMAs
    auto step_g1 = [&](f32x16 &nxt, const f32x16 &cur, int c, half8 &hfo, auto kh_c, auto act_c) {
        constexpr int kh = decltype(kh_c)::value;
        constexpr bool ACT = decltype(act_c)::value != 0;
        put(u + 2);
        fetch(u + 3);
        const float *bp = sB1 + c * 32 + 16 * kh + 4 * hi;
        QV_SB();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // fragment i of this unit is in buffer i % 4 (read four MFMAs ago); its buffer is refilled right behind the MFMA
            if (ABL & 2) {
                f32x16 z = {};
                const f32x16 t = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i & 3], xf[16 * kh + i], z, 0, 0, 0);
                asm volatile("" ::"v"(t));
            } else
            nxt = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i & 3], xf[16 * kh + i], nxt, 0, 0, 0);
            rd1(fr[i & 3], i < 12 ? u : u + 1, (i + 4) & 15);
            if (ACT && !(ABL & 1) && (i & 3) == 0) {
                const int e = i >> 2;
                const float a = cur[8 * kh + e] + bp[e], b = cur[8 * kh + 4 + e] + bp[8 + e];
                hfo[e] = (half_t)(a * ffn_sigm(a));
                hfo[4 + e] = (half_t)(b * ffn_sigm(b));
            }
            if ((i & 3) == 3) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    QV_SGB(SG_MFMA, 1);
                    QV_SGB(SG_DSR, 1);
                    if (ACT) QV_SGB(SG_VALU, 6);
                }
                QV_SB();
            }
        }
        step_end();
    };
    // GEMM2 step: output tiles 8 nh .. 8 nh + 7, the chunk's 32 hidden channels as two 16-deep k-steps (fragment 2 j + s)
    auto step_g2 = [&](const half8 (&hf)[2], auto nh_c) {
        constexpr int nh = decltype(nh_c)::value;
        put(u + 2);
        fetch(u + 3);
        QV_SB();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (ABL & 32) {
                f32x16 z = {};
                const f32x16 t = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i & 3], hf[i & 1], z, 0, 0, 0);
                asm volatile("" ::"v"(t));
            } else
            acc2[nh * 8 + (i >> 1)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i & 3], hf[i & 1], acc2[nh * 8 + (i >> 1)], 0, 0, 0);
            rd1(fr[i & 3], i < 12 ? u : u + 1, (i + 4) & 15);
            if ((i & 3) == 3) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    QV_SGB(SG_MFMA, 1);
                    QV_SGB(SG_DSR, 1);
                }
                QV_SB();
            }
        }
        step_end();
    };

    f32x16 accA, accB;   // GEMM1 accumulators of chunk c (even / odd): one is filled while the other is activated
    half8 hf[2];
    auto zero = [](f32x16 &a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
    };
    // chunk 0: GEMM1 only
    zero(accA);
    zero(accB);
    step_g1(accA, accB, 0, hf[0], C0, C0);
    step_g1(accA, accB, 0, hf[1], C1, C0);
    // steady state, two chunks per iteration so that the accumulator roles are static; the stream holds no GEMM1 units for
    // a 65th chunk, so the last chunk is activated without one
    auto body = [&](f32x16 &cur, f32x16 &nxt, int c) {
        zero(nxt);
        step_g1(nxt, cur, c, hf[0], C0, C1);
        step_g1(nxt, cur, c, hf[1], C1, C1);
        step_g2(hf, C0);
        step_g2(hf, C1);
    };
    for (int c = 0; c + 2 < NCHUNK; c += 2) {
        body(accA, accB, c);
        body(accB, accA, c + 1);
    }
    body(accA, accB, NCHUNK - 2);
    {   // last chunk (odd index: lives in accB)
        const float *bp = sB1 + (NCHUNK - 1) * 32 + 4 * hi;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const f32x4 ba = *(const f32x4 *)(bp + 16 * part), bb = *(const f32x4 *)(bp + 16 * part + 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) act1(accB, ba, bb, part, e, hf[part]);
        }
        step_g2(hf, C0);
        step_g2(hf, C1);
    }

    // ---- epilogue: out = out + alpha * (acc + b2); lane (token, hi), tile nt, register 4 q + e <-> column 32 nt + 8 q + 4 hi + e
    if (m0 + tok < g.M) {
        float *op = g.out + (size_t)(m0 + tok) * g.ldo + 4 * hi;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = nt * 32 + 8 * q;
                const f32x4 b = *(const f32x4 *)(g.b2 + col + 4 * hi);
                f32x4 o = *(const f32x4 *)(op + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += g.alpha * (acc2[nt][4 * q + e] + b[e]);
                *(f32x4 *)(op + col) = o;
            }
    }
}

}  // namespace

// ---- host: the unit stream (see qv_ffn.h).  Fragment f of a unit = 1 KB = lane l's 8 halves at f * 1024 + l * 16.
//   GEMM1 unit (chunk c, K half kh), fragment ks: lane l <-> W1[32 c + (l & 31)][256 kh + 16 ks + 8 (l >> 5) + i], i = 0..7
//   GEMM2 unit (chunk c, N half nh), fragment 2 j + s: lane l <-> W2[32 (8 nh + j) + (l & 31)][32 c + 16 s + kk(i)],
//     kk(i) = 4 (l >> 5) + i for i < 4, 8 + 4 (l >> 5) + (i - 4) for i >= 4   (the accumulator layout of GEMM1's output)
void qv_ffn_pack(const float *w1, const float *w2, half_t *out) {
    size_t u = 0;
    auto g1_unit = [&](int c, int kh) {
        half_t *p = out + u * (QV_FFN_UNIT_BYTES / 2);
        for (int ks = 0; ks < 16; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 8; ++i)
                    p[(ks * 64 + l) * 8 + i] = (half_t)w1[(size_t)(32 * c + (l & 31)) * QV_D + 256 * kh + 16 * ks + 8 * (l >> 5) + i];
        ++u;
    };
    auto g2_unit = [&](int c, int nh) {
        half_t *p = out + u * (QV_FFN_UNIT_BYTES / 2);
        for (int j = 0; j < 8; ++j)
            for (int s = 0; s < 2; ++s)
                for (int l = 0; l < 64; ++l)
                    for (int i = 0; i < 8; ++i) {
                        const int kk = i < 4 ? 4 * (l >> 5) + i : 8 + 4 * (l >> 5) + (i - 4);
                        p[((j * 2 + s) * 64 + l) * 8 + i] = (half_t)w2[(size_t)(32 * (8 * nh + j) + (l & 31)) * QV_FF + 32 * c + 16 * s + kk];
                    }
        ++u;
    };
    g1_unit(0, 0); g1_unit(0, 1);
    for (int c = 0; c < NCHUNK; ++c) {
        if (c + 1 < NCHUNK) { g1_unit(c + 1, 0); g1_unit(c + 1, 1); }
        g2_unit(c, 0); g2_unit(c, 1);
    }
}

#ifdef QV_FFN_ABLATIONS
template <int ABL>
void launch_ffn_fused_abl(const FfnArgs &a, hipStream_t s) {
    (void)hipFuncSetAttribute((const void *)k_ffn_fused<ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * UNIT + QV_FF * 4);
    hipLaunchKernelGGL(k_ffn_fused<ABL>, dim3((a.M + 127) / 128), dim3(256), NSLOT * UNIT + QV_FF * 4, s, a);
}
#endif

void launch_ffn_fused(const FfnArgs &a, hipStream_t s) {
    static bool opted = false;
    if (!opted) {
        (void)hipFuncSetAttribute((const void *)k_ffn_fused<0>, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * UNIT + QV_FF * 4);
        opted = true;
    }
    hipLaunchKernelGGL(k_ffn_fused<0>, dim3((a.M + 127) / 128), dim3(256), NSLOT * UNIT + QV_FF * 4, s, a);
}
