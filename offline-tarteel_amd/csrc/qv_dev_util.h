// qv_dev_util.h -- small device helpers shared by qv_layers.hip and qv_ort.hip (one definition, so that the two
// translation units normalise a row / a mel feature with the very same float32 operations).
#pragma once

#include "qv_kernels.h"

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
static __device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

// XCD-affine work order (round 5).  The dispatcher hands consecutive workgroup ids to the 8 XCDs round-robin (id % 8), and
// every XCD has a private 4 MB L2.  The GEMM kernels give XCD k the k-th contiguous run of their tiles, i.e. the k-th eighth
// of the rows of the activation; a row-wise kernel that lets block id i work on row group i instead spreads every eighth
// over all eight L2s and finds nothing the GEMM before it wrote.  xcd_order maps a workgroup id to a position in the
// row-ordered work list such that XCD k gets the k-th contiguous chunk (bijective for any count, the GEMMs' formula):
// k_layernorm2 (reads and rewrites the residual stream the FFN-down GEMM just wrote) 14.5 -> 9.2 us per launch, k_dwconv1d
// 10.55 -> 9.72, k_layernorm 5.64 -> 5.4-5.6 (profiles/archive/r05_o_xcd_order_kernel_averages.log); the attention kernels did not gain.
static __device__ __forceinline__ int xcd_order(int wg, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// per-feature mean / reciprocal std of the log-mel features from the f64 sums k_melstats accumulated
static __device__ __forceinline__ void mel_mean_rstd(const double *acc, int b, int f, int tm, float &mean, float &rstd) {
    double s1 = acc[((size_t)b * QV_NMEL + f) * 2], s2 = acc[((size_t)b * QV_NMEL + f) * 2 + 1];
    double mu = s1 / tm, var = (s2 - s1 * mu) / (tm - 1);
    mean = (float)mu;
    rstd = 1.f / (sqrtf((float)(var > 0.0 ? var : 0.0)) + 1e-5f);
}

// LayerNorm parameters of one lane (8 channels), requested before the row statistics so that
// their latency overlaps the reductions
struct LnParam { f32x4 g0, g1, b0, b1; };
static __device__ __forceinline__ LnParam ln_param(const float *__restrict__ gam, const float *__restrict__ bet, int lane) {
    LnParam p;
    p.g0 = *(const f32x4 *)(gam + lane * 8); p.g1 = *(const f32x4 *)(gam + lane * 8 + 4);
    p.b0 = *(const f32x4 *)(bet + lane * 8); p.b1 = *(const f32x4 *)(bet + lane * 8 + 4);
    return p;
}
// one wave, one row of 512 (8 values per lane): two-pass variance in registers
static __device__ __forceinline__ void ln_row_p(const float v[8], const LnParam &p, float o[8]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    float mu = wave_sum(s) * (1.f / QV_D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { float d = v[i] - mu; q += d * d; }
    float rs = rsqrtf(wave_sum(q) * (1.f / QV_D) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = (v[i] - mu) * rs * p.g0[i] + p.b0[i]; o[4 + i] = (v[4 + i] - mu) * rs * p.g1[i] + p.b1[i]; }
}

// conv.0 (3 x 3, stride 2, pad 1 over [frames][80 bins]) on the f32 matrix pipe, v_mfma_f32_32x32x2_f32: two taps per
// instruction, five instructions (the tenth tap meets a zero weight).  The B operand of lane (l31 = position, hi = tap
// parity) for instruction j is the input sample of tap 2j + hi at that position.
// the five B operands of one position: input rows r0 .. r0 + 2 (LDS row pitch QV_NMEL + 2, column f + 1), columns 2 f1 + kx.
// koff[j] is the lane's offset of tap 2j + hi inside that 3 x 3 window (conv0_tap_offsets; the tenth tap reads the ninth's
// sample against a zero weight), so the fetch is five plain LDS reads
static __device__ __forceinline__ void conv0_tap_offsets(int hi, int koff[5]) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int k = 2 * j + hi < 9 ? 2 * j + hi : 8;
        koff[j] = (k / 3) * (QV_NMEL + 2) + k % 3;
    }
}
static __device__ __forceinline__ void conv0_taps(const float *rows, int r0, int f1, const int koff[5], float xb[5]) {
    const float *base = rows + r0 * (QV_NMEL + 2) + 2 * f1;
#pragma unroll
    for (int j = 0; j < 5; ++j) xb[j] = base[koff[j]];
}
