// qv_ort.hip -- non-GEMM kernels of QV_PREC_ORT_MIXED (see qv_ort.h): range tracking, DynamicQuantizeLinear, and the
// depthwise / strided convolutions of the model as exact integer convolutions.
//
// Integer arithmetic in float32: |x_q - zp| <= 255, |w_q| <= 127 and at most 9 taps, so every product and every
// partial sum is an integer below 2^24 -- fmaf on integer-valued floats IS the int32 accumulation of ConvInteger.
// (The GEMM-shaped convolutions run on the i8 MFMA instead: k_gemm<.., WQ = 88, ..> in qv_gemm.hip.)
// All of these are HBM-bound elementwise / small-stencil kernels; they keep the tiling of their f16 counterparts in
// qv_layers.hip (a lane owns 8 channels, weights in registers, sliding windows), only the element type changes.

#include "qv_ort.h"
#include "qv_dev_util.h"

#include <math.h>

namespace {

__global__ void k_mm_init(uint32_t *__restrict__ mm, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mm[i] = QV_MM_INIT;
}

__device__ __forceinline__ int owner_utt(const RowOwner &o, int row) { return o.row_map ? (o.row_map[row] >> 16) : row / o.rows_per_utt; }

// Folding a range into a site's pair is an L2 atomic on ONE address per utterance: thousands of waves doing it at once
// serialise there (first version: k_ln_ort 79 us instead of 6, the GLU GEMM 200 us instead of 15).  So ranges are
// reduced inside the block first -- one atomic pair per (block, utterance) -- and mm_fold's plain read drops the pair
// altogether once the site's keys have moved past it.
//
// whole block = ONE utterance: thread-private ranges (inactive threads pass +inf / -inf) -> one fold by thread 0.
// Every thread of the block must call it (it synchronises); sh: 2 floats per wave.
__device__ __forceinline__ void block_fold(uint32_t *mm, float mn, float mx, float *sh) {
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) { sh[2 * wave] = mn; sh[2 * wave + 1] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) { mn = fminf(mn, sh[2 * w]); mx = fmaxf(mx, sh[2 * w + 1]); }
        if (mn <= mx) mm_fold(mm, mn, mx);
    }
}
// rows of several utterances in one block: per-row ranges in LDS (utt < 0 = no row), rows of an utterance are
// consecutive; the first row of every run reduces its run and folds it.  Threads 0..n_rows-1 take part, after a barrier.
__device__ __forceinline__ void rows_fold(uint32_t *mm_site, const int *s_utt, const float *s_mn, const float *s_mx, int n_rows, int r) {
    if (r >= n_rows || s_utt[r] < 0) return;
    if (r > 0 && s_utt[r - 1] == s_utt[r]) return;
    float mn = s_mn[r], mx = s_mx[r];
    for (int k = r + 1; k < n_rows && s_utt[k] == s_utt[r]; ++k) { mn = fminf(mn, s_mn[k]); mx = fmaxf(mx, s_mx[k]); }
    mm_fold(mm_site + QV_MM_STRIDE * s_utt[r], mn, mx);
}

// ------------------------------------------------------------------ quantise ----------
// f32 [rows][C] -> s8 [rows][C]; a thread converts 16 consecutive values (one 16-byte store)
__global__ __launch_bounds__(256) void k_quant_rows(const float *__restrict__ x, int rows, int C, const RowOwner own,
                                                    const uint32_t *__restrict__ mm, int8_t *__restrict__ y) {
    const int cpr = C >> 4;
    const size_t idx = (size_t)xcd_order(blockIdx.x, gridDim.x) * 256 + threadIdx.x;   // rows in XCD-affine order (qv_dev_util.h)
    if (idx >= (size_t)rows * cpr) return;
    const int row = (int)(idx / cpr), c = (int)(idx - (size_t)row * cpr) << 4;
    const QParam p = dql_param(mm + QV_MM_STRIDE * owner_utt(own, row));
    const float *px = x + (size_t)row * C + c;
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *(const f32x4 *)(px + 4 * k);
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t u = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = (int)(quant_c(v[k][e], p) + (p.zp - 128.f));
            u |= ((uint32_t)q & 0xFFu) << (8 * e);
        }
        w[k] = u;
    }
    *(uint4 *)(y + (size_t)row * C + c) = make_uint4(w[0], w[1], w[2], w[3]);
}

// ------------------------------------------------------------------ LayerNorm ----------
// k_layernorm's row handling (qv_layers.hip): one wave normalises LN_ROWS consecutive rows of 512.
// QUANT = false: fold the row's output range into its utterance's pair (and keep the f32 output when y32 is set);
// QUANT = true : recompute the same values, write them as s8.
#define LNQ_ROWS 2
template <bool QUANT>
__global__ __launch_bounds__(256) void k_ln_ort(const float *__restrict__ x, const float *__restrict__ gam,
                                                const float *__restrict__ bet, int M, const int32_t *__restrict__ row_map,
                                                uint32_t *__restrict__ mm, float *__restrict__ y32, int8_t *__restrict__ y8) {
    __shared__ int s_utt[4 * LNQ_ROWS];
    __shared__ float s_mn[4 * LNQ_ROWS], s_mx[4 * LNQ_ROWS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = (xcd_order(blockIdx.x, gridDim.x) * 4 + wave) * LNQ_ROWS;   // XCD-affine row order (qv_dev_util.h)
    f32x4 a[LNQ_ROWS], c[LNQ_ROWS];
#pragma unroll
    for (int r = 0; r < LNQ_ROWS; ++r) {
        const int row = row0 + r < M ? row0 + r : M - 1;
        const float *p = x + (size_t)row * QV_D + lane * 8;
        a[r] = *(const f32x4 *)p; c[r] = *(const f32x4 *)(p + 4);
    }
    const LnParam pr = ln_param(gam, bet, lane);
#pragma unroll
    for (int r = 0; r < LNQ_ROWS; ++r) {
        const int row = row0 + r;
        if (row >= M) {
            if (!QUANT && lane == 0) s_utt[wave * LNQ_ROWS + r] = -1;
            continue;
        }
        const int utt = row_map[row] >> 16;
        float v[8] = {a[r][0], a[r][1], a[r][2], a[r][3], c[r][0], c[r][1], c[r][2], c[r][3]}, o[8];
        ln_row_p(v, pr, o);
        if (!QUANT) {
            float mn = o[0], mx = o[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) { mn = fminf(mn, o[i]); mx = fmaxf(mx, o[i]); }
            mn = wave_min(mn);
            mx = wave_max(mx);
            if (lane == 0) { s_utt[wave * LNQ_ROWS + r] = utt; s_mn[wave * LNQ_ROWS + r] = mn; s_mx[wave * LNQ_ROWS + r] = mx; }
            if (y32) {
                float *q = y32 + (size_t)row * QV_D + lane * 8;
                *(f32x4 *)q = f32x4{o[0], o[1], o[2], o[3]};
                *(f32x4 *)(q + 4) = f32x4{o[4], o[5], o[6], o[7]};
            }
        } else {
            const QParam p = dql_param(mm + QV_MM_STRIDE * utt);
            uint32_t w[2] = {0, 0};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int q = (int)(quant_c(o[i], p) + (p.zp - 128.f));
                w[i >> 2] |= ((uint32_t)q & 0xFFu) << (8 * (i & 3));
            }
            *(uint2 *)(y8 + (size_t)row * QV_D + lane * 8) = make_uint2(w[0], w[1]);
        }
    }
    if (!QUANT) {
        __syncthreads();
        rows_fold(mm, s_utt, s_mn, s_mx, 4 * LNQ_ROWS, threadIdx.x);
    }
}

// range of f32 [M][512] packed rows: a wave takes 4 rows, a block 16
#define RMM_ROWS 4
__global__ __launch_bounds__(256) void k_rows_minmax(const float *__restrict__ x, int M, const int32_t *__restrict__ row_map,
                                                     uint32_t *__restrict__ mm) {
    __shared__ int s_utt[4 * RMM_ROWS];
    __shared__ float s_mn[4 * RMM_ROWS], s_mx[4 * RMM_ROWS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < RMM_ROWS; ++r) {
        const int row = (xcd_order(blockIdx.x, gridDim.x) * 4 + wave) * RMM_ROWS + r;
        float mn = INFINITY, mx = -INFINITY;
        if (row < M) {
            const float *p = x + (size_t)row * QV_D + lane * 8;
            const f32x4 a = *(const f32x4 *)p, c = *(const f32x4 *)(p + 4);
            mn = fminf(fminf(fminf(a[0], a[1]), fminf(a[2], a[3])), fminf(fminf(c[0], c[1]), fminf(c[2], c[3])));
            mx = fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3])));
        }
        mn = wave_min(mn);
        mx = wave_max(mx);
        if (lane == 0) { s_utt[wave * RMM_ROWS + r] = row < M ? (row_map[row] >> 16) : -1; s_mn[wave * RMM_ROWS + r] = mn; s_mx[wave * RMM_ROWS + r] = mx; }
    }
    __syncthreads();
    rows_fold(mm, s_utt, s_mn, s_mx, 4 * RMM_ROWS, threadIdx.x);
}

// ------------------------------------------------------------------ conv module --------
// depthwise Conv1d(512, k = 9, pad 4) as ConvInteger on the quantised GLU output, then BatchNorm (eval) and Swish.
//   y = float(sum_k (x_q[t + k - 4] - zp) * w_q[k]) * (s_x * s_w) + bias
//   z = fma(y, alpha, beta)      alpha = gamma / sqrt(var + eps), beta = fma(-mean, alpha, bn_bias): torch's eval-mode
//                                batch_norm on the CPU, bit for bit (tests/test_oracle_ort_semantics.py)
//   out = z * sigmoid(z)
// Frames t >= len read as zero point (real 0), as the reference's padding does.  Same ownership as k_dwconv1d: a lane
// owns 8 channels and slides over DWQ_TT consecutive frames.
#define DWQ_TT 8
__global__ __launch_bounds__(256) void k_dwconv1d_ort(const float *__restrict__ x, const float *__restrict__ wq, float w_scale,
                                                      const float *__restrict__ bias, const float *__restrict__ bn_alpha,
                                                      const float *__restrict__ bn_beta, const int32_t *__restrict__ len,
                                                      const int32_t *__restrict__ row_off, const uint32_t *__restrict__ mm_in,
                                                      uint32_t *__restrict__ mm_out, float *__restrict__ y) {
    __shared__ float s_fold[8];
    const int wg = xcd_order(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);   // (utterance, chunk) order, XCD-affine
    const int b = wg / gridDim.x, bx = wg - b * gridDim.x, t0 = (bx * 4 + (threadIdx.x >> 6)) * DWQ_TT, lane = threadIdx.x & 63;
    const int T = len[b], c0 = lane * 8;
    if (bx * 4 * DWQ_TT >= T) return;   // (whole block)
    const size_t row0 = (size_t)row_off[b];
    const QParam p = dql_param(mm_in + QV_MM_STRIDE * b);
    const float sxw = p.scale * w_scale;
    float mn = INFINITY, mx = -INFINITY;
    if (t0 < T) {
    float w[9][8];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        f32x4 w0 = *(const f32x4 *)(wq + k * QV_D + c0), w1 = *(const f32x4 *)(wq + k * QV_D + c0 + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { w[k][c] = w0[c]; w[k][4 + c] = w1[c]; }
    }
    float acc[DWQ_TT][8];
#pragma unroll
    for (int j = 0; j < DWQ_TT; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[j][c] = 0.f;
#pragma unroll
    for (int i = 0; i < DWQ_TT + 8; ++i) {
        const int tt = t0 - 4 + i;
        if (tt < 0 || tt >= T) continue;
        const float *px = x + (row0 + tt) * QV_D + c0;
        const f32x4 v0 = *(const f32x4 *)px, v1 = *(const f32x4 *)(px + 4);
        float vf[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) { vf[c] = quant_c(v0[c], p); vf[4 + c] = quant_c(v1[c], p); }
#pragma unroll
        for (int j = 0; j < DWQ_TT; ++j) {
            const int k = i - j;  // tap index: tt = (t0 + j) + k - 4
            if (k < 0 || k > 8) continue;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[j][c] = __builtin_fmaf(w[k][c], vf[c], acc[j][c]);
        }
    }
    float bs[8], al[8], be[8];
    {
        f32x4 b0 = *(const f32x4 *)(bias + c0), b1 = *(const f32x4 *)(bias + c0 + 4);
        f32x4 a0 = *(const f32x4 *)(bn_alpha + c0), a1 = *(const f32x4 *)(bn_alpha + c0 + 4);
        f32x4 e0 = *(const f32x4 *)(bn_beta + c0), e1 = *(const f32x4 *)(bn_beta + c0 + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { bs[c] = b0[c]; bs[4 + c] = b1[c]; al[c] = a0[c]; al[4 + c] = a1[c]; be[c] = e0[c]; be[4 + c] = e1[c]; }
    }
#pragma unroll
    for (int j = 0; j < DWQ_TT; ++j) {
        if (t0 + j >= T) break;
        float o[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float yv = acc[j][c] * sxw + bs[c];
            const float z = __builtin_fmaf(yv, al[c], be[c]);
            o[c] = z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-(z * 0x1.715476p+0f)));   // Swish on v_exp_f32 / v_rcp_f32
            mn = fminf(mn, o[c]);
            mx = fmaxf(mx, o[c]);
        }
        float *q = y + (row0 + t0 + j) * QV_D + c0;
        *(f32x4 *)q = f32x4{o[0], o[1], o[2], o[3]};
        *(f32x4 *)(q + 4) = f32x4{o[4], o[5], o[6], o[7]};
    }
    }
    block_fold(mm_out + QV_MM_STRIDE * b, mn, mx, s_fold);
}

// ------------------------------------------------------------------ front-end ----------
// range of the normalised log-mel features of the valid frames (the input of conv.0's DynamicQuantizeLinear); the
// normalisation is the expression k_sub01_ort applies on load
#define MMQ_CHUNKS 16
__global__ __launch_bounds__(320) void k_mel_minmax(const float *__restrict__ feats, const int32_t *__restrict__ n_samples, int tm_max,
                                                    const double *__restrict__ stats, uint32_t *__restrict__ mm) {
    const int b = blockIdx.y, f = threadIdx.x % QV_NMEL, g = threadIdx.x / QV_NMEL;  // 4 time groups
    const int tm = n_samples[b] / 160 + 1;
    const int per = (tm + MMQ_CHUNKS - 1) / MMQ_CHUNKS, t0 = blockIdx.x * per, t1 = min(tm, t0 + per);
    const float *x = feats + (size_t)b * tm_max * QV_NMEL;
    float mean, rstd;
    mel_mean_rstd(stats, b, f, tm, mean, rstd);
    float mn = INFINITY, mx = -INFINITY;
    for (int t = t0 + g; t < t1; t += 4) {
        const float v = (x[t * QV_NMEL + f] - mean) * rstd;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    __shared__ float s_fold[10];
    block_fold(mm + QV_MM_STRIDE * b, mn, mx, s_fold);
}

// conv.0 (Conv2d 1 -> 256, 3x3, s2, p1) + ReLU and conv.2 (depthwise 3x3, s2, p1) as integer convolutions; k_sub01's
// tiling (qv_layers.hip): block = (64-channel group, 4 output frames, utterance), the 19 mel rows it needs sit in LDS
// -- here as x_q - zp --, the 9 x 40 x 64 conv.0 tile is computed into LDS and the depthwise conv reads it from there.
//   PASS 0: only the range of ReLU(conv.0) over the valid frames (its DynamicQuantizeLinear needs the max first);
//   PASS 1: conv.0 again, quantised with that range into the tile (integers 0..255: exact as f16), then conv.2 ->
//           f32 [B][T2][20][256] with its range folded for conv.3's quantiser.
// (Round 4 had a version of this kernel with conv.0 on the matrix pipe -- one v_mfma_f32_32x32x16_f16 per 32 positions x 32
// channels, all four channel groups in one block: range pass 715 -> 435 us, real pass 1546 -> 1379 us at B = 256, every
// integer stage still bit-identical.  It was withdrawn at the end of the round: with it on the GPU, a log-mel kernel of
// ANOTHER engine running at the same moment occasionally computed a few wrong power-spectrum bins (soak against a second
// engine: 38 of 1,500 batches with 30 s clips; tools/dev_ort_race.py shows the first differing stage, the raw features and
// that the same call repeated is right; with this version as the only difference in the background engine: 0 of 300).
// The kernel had no out-of-range access in its ISA (LDS and global checked instruction by instruction); tools/
// interference_probe.hip reproduces the disturbance in a second and its ablations say it takes that kernel's f16 MFMA (a f32
// MFMA or VALU arithmetic in its place: nothing).  The proven kernel is back; DESIGN.md section 4, "Precision 2's extra".)
#define SQ_TT 4
#define SQ_R1 (2 * SQ_TT + 1)
#define SQ_RM (2 * SQ_R1 + 1)
#define SQ_CG 64
template <int PASS>
__global__ __launch_bounds__(256) void k_sub01_ort(const float *__restrict__ feats, int tm_max, const int32_t *__restrict__ len_mel,
                                                   const double *__restrict__ stats, const float *__restrict__ w0q, float w0_scale,
                                                   const float *__restrict__ b0, const int32_t *__restrict__ len1,
                                                   const float *__restrict__ w1q, float w1_scale, const float *__restrict__ b1,
                                                   const int32_t *__restrict__ len2, const uint32_t *__restrict__ mm_mel,
                                                   uint32_t *__restrict__ mm_c0, uint32_t *__restrict__ mm_c1,
                                                   float *__restrict__ out, int t2_max) {
    __shared__ float rows[SQ_RM][QV_NMEL + 2];
    __shared__ float mean_s[QV_NMEL], rstd_s[QV_NMEL];
    __shared__ __attribute__((aligned(16))) half_t tile[SQ_R1][40][SQ_CG];
    __shared__ float s_fold[8];
    const int b = blockIdx.z, t2_0 = blockIdx.y * SQ_TT, cg = blockIdx.x * SQ_CG, tid = threadIdx.x, lane = tid & 63;
    const int tin = len_mel[b], l1 = len1[b];
    const float *x = feats + (size_t)b * tm_max * QV_NMEL;
    if (tid < QV_NMEL) mel_mean_rstd(stats, b, tid, tin, mean_s[tid], rstd_s[tid]);
    __syncthreads();
    const QParam pm = dql_param(mm_mel + QV_MM_STRIDE * b);
    const int t1_0 = 2 * t2_0 - 1;         // first conv.0 row of the tile
    const int tm_0 = 2 * t1_0 - 1;         // first mel row
    for (int i = tid; i < SQ_RM * (QV_NMEL + 2); i += 256) {
        int r = i / (QV_NMEL + 2), f = i % (QV_NMEL + 2) - 1, t = tm_0 + r;
        // frames past the utterance and the conv padding are real zeros = the zero point
        rows[r][f + 1] = (t >= 0 && t < tin && f >= 0 && f < QV_NMEL) ? quant_c((x[t * QV_NMEL + f] - mean_s[f]) * rstd_s[f], pm) : 0.f;
    }
    const int c8 = (tid & 7) * 8, pl = tid >> 3;   // 8 channels per thread, 32 positions per pass
    float w[9][8], bs[8];
    auto load_w = [&](const float *wt, const float *bias) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            f32x4 wa = *(const f32x4 *)(wt + k * QV_SUBC + cg + c8), wb = *(const f32x4 *)(wt + k * QV_SUBC + cg + c8 + 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) { w[k][c] = wa[c]; w[k][4 + c] = wb[c]; }
        }
        f32x4 ba = *(const f32x4 *)(bias + cg + c8), bb = *(const f32x4 *)(bias + cg + c8 + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { bs[c] = ba[c]; bs[4 + c] = bb[c]; }
    };
    load_w(w0q, b0);
    __syncthreads();
    const float s0 = pm.scale * w0_scale;
    QParam p0 = {1.f, 0.f, 1.f};
    if (PASS == 1) p0 = dql_param(mm_c0 + QV_MM_STRIDE * b);
    float mn = INFINITY, mx = -INFINITY;
    // ---- conv.0 + ReLU (rows outside [0, l1) are the depthwise conv's zero padding)
    for (int p = pl; p < SQ_R1 * 40; p += 32) {
        int r = p / 40, f1 = p - r * 40, t1 = t1_0 + r;
        half8 o;
        if (t1 < 0 || t1 >= l1) {
#pragma unroll
            for (int c = 0; c < 8; ++c) o[c] = (half_t)0.f;
        } else {
            float acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int df = 0; df < 3; ++df) {
                    float v = rows[2 * r + dt][2 * f1 + df];
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c] = __builtin_fmaf(w[dt * 3 + df][c], v, acc[c]);
                }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float yv = acc[c] * s0 + bs[c];
                yv = yv > 0.f ? yv : 0.f;
                if (PASS == 0) { mn = fminf(mn, yv); mx = fmaxf(mx, yv); }
                else o[c] = (half_t)quant_c(yv, p0);
            }
        }
        if (PASS == 1) *(half8 *)&tile[r][f1][c8] = o;
    }
    if (PASS == 0) {
        block_fold(mm_c0 + QV_MM_STRIDE * b, mn, mx, s_fold);
        return;
    }
    load_w(w1q, b1);
    __syncthreads();
    const float s1 = p0.scale * w1_scale;
    const int l2 = len2[b];
    // ---- depthwise 3x3 stride 2 over the tile
    for (int p = pl; p < SQ_TT * 20; p += 32) {
        int tl = p / 20, fo = p - tl * 20, t2 = t2_0 + tl;
        if (t2 >= t2_max) continue;
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int df = 0; df < 3; ++df) {
                int f = 2 * fo - 1 + df;
                if (f < 0 || f >= 40) continue;
                half8 v = *(const half8 *)&tile[2 * tl + dt][f][c8];
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = __builtin_fmaf(w[dt * 3 + df][c], (float)v[c], acc[c]);
            }
        float o[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            o[c] = acc[c] * s1 + bs[c];
            if (t2 < l2) { mn = fminf(mn, o[c]); mx = fmaxf(mx, o[c]); }
        }
        float *q = out + (((size_t)b * t2_max + t2) * 20 + fo) * QV_SUBC + cg + c8;
        *(f32x4 *)q = f32x4{o[0], o[1], o[2], o[3]};
        *(f32x4 *)(q + 4) = f32x4{o[4], o[5], o[6], o[7]};
    }
    block_fold(mm_c1 + QV_MM_STRIDE * b, mn, mx, s_fold);
}

// Round 5: conv.0 on the FLOAT32 matrix pipe, all four channel groups in one block (the re-landed form of the withdrawn
// kernel above: tools/withdrawn/qv_ort_conv0_mfma.hip with PROBE_ABL=8192, which leaves a co-running k_logmel undisturbed
// in tools/interference_probe -- tests/test_gpu_interference.py runs that probe).  [channels] x [9 taps] x [positions] as
// five v_mfma_f32_32x32x2_f32 per 32 x 32 tile like k_sub01 (qv_layers.hip), on the integer-valued operands x_q - zp and
// w_q: every product and partial sum is an integer below 2^24, so the instruction's internal summation order cannot
// change a bit and the VALU kernel above stays the bit-exact cross-check (QVERSE_ORT_SUB=0 /
// qv_debug_kernel_variant(1, 0); tests/test_gpu_ort_mixed.py compares the two and both with the oracle).
// One block walks the four 64-channel groups over the same 19 quantised mel rows: with a block per group the per-block
// fixed work (the f64 feature statistics, the quantisation of the rows, two barriers) was most of the kernel.
#define SQ_NPOS (SQ_R1 * 40)     // conv.0 positions of a tile
template <int PASS>
__global__ __launch_bounds__(256) void k_sub01_ort_mx(const float *__restrict__ feats, int tm_max, const int32_t *__restrict__ len_mel,
                                                      const double *__restrict__ stats, const float *__restrict__ w0q, float w0_scale,
                                                      const float *__restrict__ b0, const int32_t *__restrict__ len1,
                                                      const float *__restrict__ w1q, float w1_scale, const float *__restrict__ b1,
                                                      const int32_t *__restrict__ len2, const uint32_t *__restrict__ mm_mel,
                                                      uint32_t *__restrict__ mm_c0, uint32_t *__restrict__ mm_c1,
                                                      float *__restrict__ out, int t2_max) {
    __shared__ float rows[SQ_RM][QV_NMEL + 2];
    __shared__ float mean_s[QV_NMEL], rstd_s[QV_NMEL];
    // tile[position][64 channels]: the 16-byte chunk c of a position sits at (c + position) & 7, so that the 32 positions
    // of a wave store spread over the banks while the depthwise conv's 8 threads per position still read one contiguous 128 bytes
    __shared__ __attribute__((aligned(16))) half_t tile[PASS == 1 ? SQ_NPOS : 1][SQ_CG];
    __shared__ float s_fold[8];
    const int b = blockIdx.z, t2_0 = blockIdx.y * SQ_TT, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tin = len_mel[b], l1 = len1[b];
    const float *x = feats + (size_t)b * tm_max * QV_NMEL;
    if (tid < QV_NMEL) mel_mean_rstd(stats, b, tid, tin, mean_s[tid], rstd_s[tid]);
    __syncthreads();
    const QParam pm = dql_param(mm_mel + QV_MM_STRIDE * b);
    const int t1_0 = 2 * t2_0 - 1;         // first conv.0 row of the tile
    const int tm_0 = 2 * t1_0 - 1;         // first mel row
    for (int i = tid; i < SQ_RM * (QV_NMEL + 2); i += 256) {
        int r = i / (QV_NMEL + 2), f = i % (QV_NMEL + 2) - 1, t = tm_0 + r;
        // frames past the utterance and the conv padding are real zeros = the zero point
        rows[r][f + 1] = (t >= 0 && t < tin && f >= 0 && f < QV_NMEL) ? quant_c((x[t * QV_NMEL + f] - mean_s[f]) * rstd_s[f], pm) : 0.f;
    }
    const int l31 = lane & 31, hi = lane >> 5;
    int koff[5];
    conv0_tap_offsets(hi, koff);
    const float s0 = pm.scale * w0_scale;
    QParam p0 = {1.f, 0.f, 1.f};
    if (PASS == 1) p0 = dql_param(mm_c0 + QV_MM_STRIDE * b);
    const float s1 = p0.scale * w1_scale;
    const int l2 = len2[b];
    float mn = INFINITY, mx = -INFINITY;     // PASS 0: range of ReLU(conv.0); PASS 1: range of conv.2
    __syncthreads();
    for (int cg = 0; cg < QV_SUBC; cg += SQ_CG) {
        // ---- conv.0 weights of this group's two 32-channel tiles as A operands: lane (l31 = channel, hi) holds tap 2j + hi
        float wa[2][5];
        f32x4 bia[2][4];                     // bias of the 16 channels this lane owns per tile: channel 8 q + 4 hi + e
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int j = 0; j < 5; ++j) wa[ct][j] = 2 * j + hi < 9 ? w0q[(2 * j + hi) * QV_SUBC + cg + ct * 32 + l31] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) bia[ct][q] = *(const f32x4 *)(b0 + cg + ct * 32 + 8 * q + 4 * hi);
        }
        // ---- conv.0 + ReLU: 12 position tiles of 32 (360 positions), three per wave
        for (int pt = wave; pt * 32 < SQ_NPOS; pt += 4) {
            const int p = pt * 32 + l31;
            const bool inside = p < SQ_NPOS;
            const int pc = inside ? p : SQ_NPOS - 1;
            const int r = pc / 40, f1 = pc - r * 40;
            float xb[5];
            conv0_taps(&rows[0][0], 2 * r, f1, koff, xb);
            const int t1 = t1_0 + r;
            const bool live = inside && t1 >= 0 && t1 < l1;    // rows outside [0, l1) are the depthwise conv's zero padding
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                f32x16 acc = {};
#pragma unroll
                for (int j = 0; j < 5; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ct][j], xb[j], acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    half4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float yv = acc[4 * q + e] * s0 + bia[ct][q][e];
                        yv = yv > 0.f ? yv : 0.f;
                        if (PASS == 0) { if (live) { mn = fminf(mn, yv); mx = fmaxf(mx, yv); } }
                        else o[e] = (half_t)(live ? quant_c(yv, p0) : 0.f);
                    }
                    if (PASS == 1 && inside) {
                        const int c = ct * 4 + q;
                        *(half4 *)&tile[p][(((c + p) & 7) << 3) + 4 * hi] = o;
                    }
                }
            }
        }
        if (PASS == 0) continue;
        const int c8 = (tid & 7) * 8, pl = tid >> 3;   // depthwise conv: 8 channels per thread, 32 positions per pass
        float w[9][8], bs[8];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            f32x4 wa4 = *(const f32x4 *)(w1q + k * QV_SUBC + cg + c8), wb4 = *(const f32x4 *)(w1q + k * QV_SUBC + cg + c8 + 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) { w[k][c] = wa4[c]; w[k][4 + c] = wb4[c]; }
        }
        {
            f32x4 ba = *(const f32x4 *)(b1 + cg + c8), bb = *(const f32x4 *)(b1 + cg + c8 + 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) { bs[c] = ba[c]; bs[4 + c] = bb[c]; }
        }
        __syncthreads();
        // ---- depthwise 3x3 stride 2 over the tile
        for (int p = pl; p < SQ_TT * 20; p += 32) {
            int tl = p / 20, fo = p - tl * 20, t2 = t2_0 + tl;
            if (t2 >= t2_max) continue;
            float acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int df = 0; df < 3; ++df) {
                    int f = 2 * fo - 1 + df;
                    if (f < 0 || f >= 40) continue;
                    const int pos = (2 * tl + dt) * 40 + f;
                    half8 v = *(const half8 *)&tile[pos][(((c8 >> 3) + pos) & 7) << 3];
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c] = __builtin_fmaf(w[dt * 3 + df][c], (float)v[c], acc[c]);
                }
            float o[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                o[c] = acc[c] * s1 + bs[c];
                if (t2 < l2) { mn = fminf(mn, o[c]); mx = fmaxf(mx, o[c]); }
            }
            float *q = out + (((size_t)b * t2_max + t2) * 20 + fo) * QV_SUBC + cg + c8;
            *(f32x4 *)q = f32x4{o[0], o[1], o[2], o[3]};
            *(f32x4 *)(q + 4) = f32x4{o[4], o[5], o[6], o[7]};
        }
        __syncthreads();                       // the tile is rewritten by the next channel group
    }
    block_fold((PASS == 0 ? mm_c0 : mm_c1) + QV_MM_STRIDE * b, mn, mx, s_fold);
}

// conv.5: depthwise Conv2d(256, 3x3, s2, p1) as ConvInteger on the quantised f32 input (channels-last); rows
// t >= len_in[b] read as the zero point.  Same ownership as k_dwconv2d: 8 channels per thread.
#define DWQ2_TT 4
__global__ __launch_bounds__(256) void k_dwconv2d_ort(const float *__restrict__ in, int tin_max, int fin,
                                                      const int32_t *__restrict__ len_in, const float *__restrict__ wq, float w_scale,
                                                      const float *__restrict__ bias, const int32_t *__restrict__ len_out,
                                                      const uint32_t *__restrict__ mm_in, uint32_t *__restrict__ mm_out,
                                                      float *__restrict__ out, int tout_max, int fout) {
    __shared__ float s_fold[8];
    const int b = blockIdx.z, to0 = blockIdx.y * DWQ2_TT, tid = threadIdx.x;   // DWQ2_TT output frames per block (see k_dwconv2d)
    const int tin = len_in[b], lout = len_out[b];
    const int c0 = (tid & 31) * 8, fl = tid >> 5;
    const QParam p = dql_param(mm_in + QV_MM_STRIDE * b);
    const float sxw = p.scale * w_scale;
    float w[9][8], bs[8];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        f32x4 w0 = *(const f32x4 *)(wq + k * QV_SUBC + c0), w1 = *(const f32x4 *)(wq + k * QV_SUBC + c0 + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { w[k][c] = w0[c]; w[k][4 + c] = w1[c]; }
    }
    {
        f32x4 b0 = *(const f32x4 *)(bias + c0), b1 = *(const f32x4 *)(bias + c0 + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { bs[c] = b0[c]; bs[4 + c] = b1[c]; }
    }
    float mn = INFINITY, mx = -INFINITY;
    for (int pp = fl; pp < DWQ2_TT * fout; pp += 8) {
        const int to = to0 + pp / fout, fo = pp % fout;
        if (to >= tout_max) break;
        const bool valid = to < lout;
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            int t = 2 * to - 1 + dt;
            if (t < 0 || t >= tin) continue;
#pragma unroll
            for (int df = 0; df < 3; ++df) {
                int f = 2 * fo - 1 + df;
                if (f < 0 || f >= fin) continue;
                const float *px = in + (((size_t)b * tin_max + t) * fin + f) * QV_SUBC + c0;
                const f32x4 v0 = *(const f32x4 *)px, v1 = *(const f32x4 *)(px + 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[c] = __builtin_fmaf(w[dt * 3 + df][c], quant_c(v0[c], p), acc[c]);
                    acc[4 + c] = __builtin_fmaf(w[dt * 3 + df][4 + c], quant_c(v1[c], p), acc[4 + c]);
                }
            }
        }
        float o[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            o[c] = acc[c] * sxw + bs[c];
            if (valid) { mn = fminf(mn, o[c]); mx = fmaxf(mx, o[c]); }
        }
        float *q = out + (((size_t)b * tout_max + to) * fout + fo) * QV_SUBC + c0;
        *(f32x4 *)q = f32x4{o[0], o[1], o[2], o[3]};
        *(f32x4 *)(q + 4) = f32x4{o[4], o[5], o[6], o[7]};
    }
    block_fold(mm_out + QV_MM_STRIDE * b, mn, mx, s_fold);
}

}  // namespace

// ====================================================================== launchers ======

void launch_mm_init(uint32_t *mm, size_t n_keys, hipStream_t s) {
    hipLaunchKernelGGL(k_mm_init, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, s, mm, n_keys);
}

void launch_quant_rows(const float *x, int rows, int C, const RowOwner &own, const uint32_t *mm, int8_t *y, hipStream_t s) {
    const size_t n = (size_t)rows * (C / 16);
    hipLaunchKernelGGL(k_quant_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, rows, C, own, mm, y);
}

void launch_ln_minmax(const float *x, const float *g, const float *b, int M, const int32_t *row_map, uint32_t *mm, float *y32,
                      hipStream_t s) {
    hipLaunchKernelGGL((k_ln_ort<false>), dim3((M + 4 * LNQ_ROWS - 1) / (4 * LNQ_ROWS)), dim3(256), 0, s, x, g, b, M, row_map, mm, y32,
                       (int8_t *)nullptr);
}

void launch_ln_quant(const float *x, const float *g, const float *b, int M, const int32_t *row_map, const uint32_t *mm, int8_t *y,
                     hipStream_t s) {
    hipLaunchKernelGGL((k_ln_ort<true>), dim3((M + 4 * LNQ_ROWS - 1) / (4 * LNQ_ROWS)), dim3(256), 0, s, x, g, b, M, row_map,
                       (uint32_t *)mm, (float *)nullptr, y);
}

void launch_rows_minmax(const float *x, int M, const int32_t *row_map, uint32_t *mm, hipStream_t s) {
    hipLaunchKernelGGL(k_rows_minmax, dim3((M + 4 * RMM_ROWS - 1) / (4 * RMM_ROWS)), dim3(256), 0, s, x, M, row_map, mm);
}

void launch_dwconv1d_ort(const float *x, const float *wq, float w_scale, const float *bias, const float *bn_alpha,
                         const float *bn_beta, const int32_t *len, const int32_t *row_off, const uint32_t *mm_in, uint32_t *mm_out,
                         float *y, int t_max, int batch, hipStream_t s) {
    hipLaunchKernelGGL(k_dwconv1d_ort, dim3((t_max + 4 * DWQ_TT - 1) / (4 * DWQ_TT), batch), dim3(256), 0, s, x, wq, w_scale, bias,
                       bn_alpha, bn_beta, len, row_off, mm_in, mm_out, y);
}

void launch_mel_minmax(const float *feats, const int32_t *n_samples, int tm_max, const double *stats, uint32_t *mm, int batch,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_mel_minmax, dim3(MMQ_CHUNKS, batch), dim3(320), 0, s, feats, n_samples, tm_max, stats, mm);
}

static int ort_sub_variant() { return qv_kernel_variant(QV_KV_ORT_SUB); }

void launch_sub01_ort(int pass, const float *feats, int tm_max, const int32_t *len_mel, const double *stats, const float *w0q,
                      float w0_scale, const float *b0, const int32_t *len1, const float *w1q, float w1_scale, const float *b1,
                      const int32_t *len2, const uint32_t *mm_mel, uint32_t *mm_c0, uint32_t *mm_c1, float *out, int t2_max,
                      int batch, hipStream_t s) {
    if (ort_sub_variant() == 1) {     // conv.0 on the f32 matrix pipe, four channel groups per block
        const dim3 g1(1, (t2_max + SQ_TT - 1) / SQ_TT, batch);
        if (pass == 0)
            hipLaunchKernelGGL((k_sub01_ort_mx<0>), g1, dim3(256), 0, s, feats, tm_max, len_mel, stats, w0q, w0_scale, b0, len1, w1q,
                               w1_scale, b1, len2, mm_mel, mm_c0, mm_c1, out, t2_max);
        else
            hipLaunchKernelGGL((k_sub01_ort_mx<1>), g1, dim3(256), 0, s, feats, tm_max, len_mel, stats, w0q, w0_scale, b0, len1, w1q,
                               w1_scale, b1, len2, mm_mel, mm_c0, mm_c1, out, t2_max);
        return;
    }
    const dim3 grid(QV_SUBC / SQ_CG, (t2_max + SQ_TT - 1) / SQ_TT, batch);
    if (pass == 0)
        hipLaunchKernelGGL((k_sub01_ort<0>), grid, dim3(256), 0, s, feats, tm_max, len_mel, stats, w0q, w0_scale, b0, len1, w1q,
                           w1_scale, b1, len2, mm_mel, mm_c0, mm_c1, out, t2_max);
    else
        hipLaunchKernelGGL((k_sub01_ort<1>), grid, dim3(256), 0, s, feats, tm_max, len_mel, stats, w0q, w0_scale, b0, len1, w1q,
                           w1_scale, b1, len2, mm_mel, mm_c0, mm_c1, out, t2_max);
}

void launch_dwconv2d_ort(const float *in, int tin_max, int fin, const int32_t *len_in, const float *wq, float w_scale,
                         const float *bias, const int32_t *len_out, const uint32_t *mm_in, uint32_t *mm_out, float *out,
                         int tout_max, int fout, int batch, hipStream_t s) {
    hipLaunchKernelGGL(k_dwconv2d_ort, dim3(1, (tout_max + DWQ2_TT - 1) / DWQ2_TT, batch), dim3(256), 0, s, in, tin_max, fin, len_in, wq, w_scale, bias,
                       len_out, mm_in, mm_out, out, tout_max, fout);
}
