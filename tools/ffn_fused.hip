// ffn_fused.hip -- fused Conformer feed-forward module: the MEASURED PROTOTYPE of VERDICT r3 item 3 (dev tool, not linked
// into libqverse.so; harness: tools/ffn_fused_bench.hip, logs: profiles/r04_e_ffn_fused_*.log).
//
// OUTCOME (round 4, one MI355X): bit-level agreement with the two-kernel path (max |d| 4.8e-7 over all outputs), but it
// LOSES: 188 us against 162 us (k_gemm256<f16_swish> + k_gemm256<resid> back to back) at M = 32,256 (252 blocks, whole
// chip), 137 us on 63 CUs against 51 us on the whole chip at M = 8,064 (8,600 against 8,000-13,000 CU-microseconds).
// Where the time goes (ablations of the harness): the MFMA streams themselves run at the full rate (64 us of 188: both
// the 16-deep accumulate chain of GEMM1 and GEMM2's independent tiles issue every 32 cycles, tools/mfma_class_probe.hip);
// launch + prologue + the 512 KB read-modify-write epilogue per block are 32 us; everything else -- 16 fragment reads of
// 1 KB, 4 staging loads + ds_writes, 4 Swish evaluations and half a barrier per 16 MFMAs and wave -- takes 53 us ALONE
// (LDS array: 4 waves x (16 reads x 4 + 4 writes x 8) = 384 of the 512 cycles a step's MFMAs last) and, with one wave per
// SIMD issuing in order, does not hide under the MFMAs but adds to them.  A wave that owns 32 tokens reads every weight
// fragment for ONE MFMA (1 KB of LDS per MFMA; the 256 x 256 GEMM tile reads 0.75 KB and has a second wave per SIMD to
// cover the waits); owning 64 tokens needs 512 accumulator registers for the output alone.  The module's working set
// (X 128 KB + output 256 KB per 128 tokens) does not leave room for a better shape on this CU -- DESIGN.md "Fused FFN".
//
// Original design notes:
//
//     fused Conformer feed-forward module
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

#include "ffn_fused.h"
#include "../offline-tarteel_amd/csrc/qv_dev_util.h"

#include <stdio.h>

#include <type_traits>
#include <vector>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NSLOT = 8;                      // LDS ring slots of one unit each (128 KB)
constexpr int UNIT = QV_FFN_UNIT_BYTES;
constexpr int NCHUNK = QV_FF / 32;            // 64 chunks of 32 hidden channels

__device__ __forceinline__ float ffn_sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

#define QV_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef QV_ACC1_AGPR
#define QV_ACC1_AGPR 1   // register class of GEMM1's accumulators: 1 = AGPR (default), 0 = VGPR (A/B experiments)
#endif
#ifndef QV_ACC2_AGPR_TILES
#define QV_ACC2_AGPR_TILES (QV_ACC1_AGPR ? 13 : 16)
#endif
#if QV_ACC1_AGPR
#define QV_ACC1_CLASS(x) "a"(x)
#else
#define QV_ACC1_CLASS(x) "v"(x)
#endif
#ifndef QV_FFN_PIN
#define QV_FFN_PIN 3     // a scheduling barrier behind every (QV_FFN_PIN + 1)-th MFMA slot: 0 = every slot, 1, 3
#endif

// ABL: timing ablations for tools/ffn_fused_bench.hip (results are WRONG with any bit set; the product launches ABL = 0):
//   1 no activation VALU, 4 no s_barrier, 8 no fragment reads, 16 no weight staging (no global loads, no ds_writes),
//   64 no GEMM1 MFMAs, 128 no GEMM2 MFMAs
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
        if (ABL & 16) { if (u > 4) return; }
#pragma unroll
        for (int i = 0; i < 4; ++i) st[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, lofs + i * 1024, u * UNIT, 0);
    };
    auto put = [&](int u) {
        if (ABL & 16) { if (u > 4) return; }
        unsigned char *s = smem + (u & (NSLOT - 1)) * UNIT + lofs;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x4 *)(s + i * 1024) = st[i];
    };
    {   // units 0..3 in one round trip (the accumulators are not live yet: registers to spare)
        u32x4 p[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) p[k][i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, lofs + i * 1024, k * UNIT, 0);
        for (int i = tid; i < QV_FF / 4; i += 256) ((f32x4 *)sB1)[i] = ((const f32x4 *)g.b1)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) *(u32x4 *)(smem + k * UNIT + lofs + i * 1024) = p[k][i];
    }
    fetch(4);

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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // One step = one unit = 16 MFMAs; one s_barrier per TWO steps.  When step u starts, units u .. u + 3 (u even) are
    // published; the step writes unit u + 4 (in the staging registers since the step before) into the ring, requests unit
    // u + 5, and reads fragments 4..15 of unit u plus 0..3 of unit u + 1 into four rotating buffers, each four MFMAs ahead of
    // its use: the LDS latency of a unit's first fragments is paid under the previous unit's MFMAs, not behind a barrier.
    int u = 0;
    half8 fr[4];
    auto rd1 = [&](half8 &f, int unit, int fi) {
        if (ABL & 8) { if (unit > 0) return; }
        f = *(const half8 *)(smem + (unit & (NSLOT - 1)) * UNIT + lane * 16 + fi * 1024);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) rd1(fr[i], 0, i);
    auto step_top = [&]() {
        put(u + 4);
        fetch(u + 5);
    };
    auto step_end = [&]() {
        if (u & 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
        }
        ++u;
    };
    const std::integral_constant<int, 0> C0{};
    const std::integral_constant<int, 1> C1{};
    const std::integral_constant<int, 2> C2{};
    const std::integral_constant<int, 3> C3{};

    // ---- activation: bias + Swish + f16 of FOUR accumulator registers (quad q of a finished GEMM1 tile) per step, cut
    // into 16 slices -- one per MFMA slot, at most one transcendental each -- so that the VALU work hides under the matrix
    // pipe instead of stalling it (one wave per SIMD: nobody else fills the gaps).  Register 4 q + e of lane (token, hi)
    // is hidden channel 8 q + 4 hi + e of the chunk and becomes k-slot 4 (q & 1) + e of GEMM2's k-step q >> 1.
    // Two values are in flight at a time (slots 0..7: elements 0, 1 of the quad; slots 8..15: elements 2, 3) -- the
    // kernel runs at the edge of the 512-register file, every live temporary counts.
    float at[2], ax[2];      // the biased values, and exp / reciprocal in flight
    auto act_slice = [&](const f32x16 &acc, const f32x4 &bias, half8 &out, auto q_c, int slot) {
        constexpr int q = decltype(q_c)::value;
        if (ABL & 1) return;
        const int v = slot & 1, e = 2 * (slot >> 3) + v;
        switch ((slot >> 1) & 3) {
            case 0: {
                // The element is fetched by an ORDERED asm read: the compiler does not know that the asm MFMAs are MFMAs, and a
                // plain `acc[k]` read let it fold both GEMM1 tiles onto one AGPR tile with copies whose placement returned
                // wrong values (round 4, max |d| 1.8e-2).  The "a" operand costs one AGPR of its own for the extracted
                // element -- which is why only 13 output tiles are pinned to AGPRs (13 x 16 + 2 x 16 + 1 <= 256).
                float a;
                if (QV_ACC1_AGPR) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a) : "a"(acc[4 * q + e]));
                else a = acc[4 * q + e];
                at[v] = a + bias[e];
                ax[v] = at[v] * -1.4426950408889634f;
                break;
            }
            case 1: ax[v] = __builtin_amdgcn_exp2f(ax[v]); break;
            case 2: ax[v] = __builtin_amdgcn_rcpf(1.0f + ax[v]); break;
            default: out[4 * (q & 1) + e] = (half_t)(at[v] * ax[v]); break;
        }
    };
    // GEMM1 step: K half kh of the NEXT chunk's hidden tile into `nxt`; quad q of `cur` is activated meanwhile
    // (`bias` = the quad's four b1 values, fetched from LDS during the previous step; this step fetches the next step's)
    f32x4 bias = {};
    auto bias_of = [&](int c, int q) { return *(const f32x4 *)(sB1 + c * 32 + 8 * q + 4 * hi); };
    auto step_g1 = [&](f32x16 &nxt, const f32x16 &cur, half8 (&hf)[2], auto kh_c, auto q_c, bool act, int c_nb, int q_nb) {
        constexpr int kh = decltype(kh_c)::value;
        constexpr int q = decltype(q_c)::value;
        step_top();
        const f32x4 bcur = bias;
        QV_SB();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // GEMM1's accumulator is pinned to ONE register class by an asm constraint.  With the builtin the allocator keeps
            // it in the AGPR half for the MFMAs and copies it out 16 registers at a time for the activation's VALU reads --
            // the duplicate tipped the kernel over the 512-register file (131 spilled VGPRs).  Measured (round 4): pinned to
            // VGPRs the 16 back-to-back dependent MFMAs of a K half issue every ~76 cycles instead of every 32 (the bare
            // MFMA stream 70 -> 118 us at M = 32,256): the accumulate chain only runs at full rate on AGPRs.  So AGPRs it
            // is, and the activation fetches single values with v_accvgpr_read (act_slice); the allocator then parks two of
            // the sixteen output tiles in VGPRs, where their once-per-16-MFMAs accumulation does not care.
            // The chunk's first MFMA takes the literal 0 as its accumulator input: no zeroing in front of it.
            if (ABL & 64) { }
            else if (kh == 0 && i == 0)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&" QV_ACC1_CLASS(nxt) : "v"(fr[i & 3]), "v"(xf[16 * kh + i]));
            else
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+" QV_ACC1_CLASS(nxt) : "v"(fr[i & 3]), "v"(xf[16 * kh + i]));
            rd1(fr[i & 3], i < 12 ? u : u + 1, (i + 4) & 15);
            if (i == 10) bias = bias_of(c_nb, q_nb);
            if (act) act_slice(cur, bcur, hf[q >> 1], q_c, i);
            if ((i & QV_FFN_PIN) == QV_FFN_PIN) QV_SB();
        }
        step_end();
    };
    // GEMM2 step: k-step ss of the chunk (16 hidden channels) into all 16 output tiles (fragment nt); quad q of `src`
    // is activated meanwhile
    auto step_g2 = [&](const half8 &hk, const f32x16 &src, half8 (&hf)[2], auto q_c, bool act, int c_nb, int q_nb) {
        constexpr int q = decltype(q_c)::value;
        step_top();
        const f32x4 bcur = bias;
        QV_SB();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // every accumulator's register class is pinned (see step_g1): GEMM1's two tiles + output tiles 0..12 + one scratch
            // element take 241 AGPRs, output tiles 13..15 live in VGPRs (tools/mfma_class_probe.hip: the class costs nothing)
            if (ABL & 128) { }
            else if (i < QV_ACC2_AGPR_TILES) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc2[i]) : "v"(fr[i & 3]), "v"(hk));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc2[i]) : "v"(fr[i & 3]), "v"(hk));
            rd1(fr[i & 3], i < 12 ? u : u + 1, (i + 4) & 15);
            if (i == 10) bias = bias_of(c_nb, q_nb);
            if (act) act_slice(src, bcur, hf[q >> 1], q_c, i);
            if ((i & QV_FFN_PIN) == QV_FFN_PIN) QV_SB();
        }
        step_end();
    };

    f32x16 accA, accB;   // GEMM1 accumulators of chunk c (even / odd): one is filled while the other is activated
    half8 hf[2];         // GEMM2 operand of the current chunk: k-steps 0 and 1
    auto act_quad = [&](const f32x16 &acc, int c, auto q_c) {   // un-overlapped (prologue / last chunk only)
        constexpr int q = decltype(q_c)::value;
        const f32x4 b = bias_of(c, q);
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) act_slice(acc, b, hf[q >> 1], q_c, sl);
    };
    // chunk 0: GEMM1 only, then its first quad
#pragma unroll
    for (int r = 0; r < 16; ++r) accB[r] = 0.f;
    step_g1(accA, accB, hf, C0, C0, false, 0, 0);
    step_g1(accA, accB, hf, C1, C0, false, 0, 1);
    act_quad(accA, 0, C0);
    // Steady state for chunk c (cur = its finished GEMM1 tile, nxt = chunk c + 1's):
    //   A  GEMM1(c+1) K half 0   | quad 1 of cur -> hf[0] k-slots 4..7
    //   B  GEMM1(c+1) K half 1   | quad 2 of cur -> hf[1] k-slots 0..3
    //   C  GEMM2(c)   k-step 0   | quad 3 of cur -> hf[1] k-slots 4..7      (hf[0] complete since A)
    //   D  GEMM2(c)   k-step 1   | quad 0 of nxt -> hf[0] k-slots 0..3      (hf[0] free since C; nxt complete since B)
    // every step carries the same VALU load; two chunks per iteration keep the accumulator roles static.
    auto body = [&](f32x16 &cur, f32x16 &nxt, int c) {
        step_g1(nxt, cur, hf, C0, C1, true, c, 2);
        step_g1(nxt, cur, hf, C1, C2, true, c, 3);
        step_g2(hf[0], cur, hf, C3, true, c + 1, 0);
        step_g2(hf[1], nxt, hf, C0, true, c + 1, 1);
    };
    for (int c = 0; c + 2 < NCHUNK; c += 2) {
        body(accA, accB, c);
        body(accB, accA, c + 1);
    }
    body(accA, accB, NCHUNK - 2);
    // last chunk (odd index: in accB, quad 0 already activated): the stream holds no GEMM1 units for a 65th chunk
    act_quad(accB, NCHUNK - 1, C1);
    act_quad(accB, NCHUNK - 1, C2);
    bias = bias_of(NCHUNK - 1, 3);
    step_g2(hf[0], accB, hf, C3, true, NCHUNK - 1, 3);
    step_g2(hf[1], accB, hf, C0, false, NCHUNK - 1, 3);

    // (the MFMAs are inline asm: the compiler does not know that the last of them are still in flight)
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
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
//   GEMM2 unit (chunk c, k-step s), fragment nt (output tile): lane l <-> W2[32 nt + (l & 31)][32 c + 16 s + kk(i)],
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
    auto g2_unit = [&](int c, int ss) {
        half_t *p = out + u * (QV_FFN_UNIT_BYTES / 2);
        for (int nt = 0; nt < 16; ++nt)
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 8; ++i) {
                    const int kk = i < 4 ? 4 * (l >> 5) + i : 8 + 4 * (l >> 5) + (i - 4);
                    p[(nt * 64 + l) * 8 + i] = (half_t)w2[(size_t)(32 * nt + (l & 31)) * QV_FF + 32 * c + 16 * ss + kk];
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
