// qv_ort.h -- QV_PREC_ORT_MIXED: the arithmetic onnxruntime runs on the reference's model file
// (experiments/c2c-direct-mixed/run.py:1-9: MatMulNBits int4 on the Linear layers, quantize_dynamic QInt8 on what
// remains, i.e. DynamicQuantizeLinear -> ConvInteger -> Cast -> Mul on EVERY Conv), restated for gfx950.
//
// Per Conv call of the reference (= per utterance, it feeds batch 1):
//   x_q   = saturate_u8(rne(x / s_x) + zp_x),  s_x = (max(x, 0) - min(x, 0)) / 255,  zp_x = saturate(rne(-min / s_x))
//   w_q   = one symmetric int8 scale per weight tensor, s_w = max|w| / 127
//   y     = float(sum (x_q - zp_x) * w_q) * (s_x * s_w) + bias          (int32 accumulation, exact)
// All in float32 exactly as oracle/fastconformer_ref.py::dynamic_quantize_linear / OrtMixed.conv evaluate it, so that
// identical inputs give identical integers and -- for the pointwise and depthwise convolutions -- identical float32
// outputs.  Activations are stored as s8 = x_q - 128 (the i8 MFMA is signed x signed); the GEMM epilogue adds
// (128 - zp_x) * sum_k w_q[n][k] to the int32 accumulator, which restores sum (x_q - zp_x) * w_q exactly.
//
// Range tracking: every quantiser site owns one {min, max} pair per utterance, kept as order-preserving u32 keys so
// that atomicMin / atomicMax do the reduction; producers fold their outputs in, consumers read the pair back.  Both
// keys start at fenc(+0.0f): DynamicQuantizeLinear widens the range to include 0.
#pragma once

#include "qv_kernels.h"

#define QV_MM_INIT 0x80000000u   // fenc(+0.0f)
// u32 slots per utterance in a site's key array: {min, max} at the front of a 256-byte slot of its own.  Packed two per
// 8 bytes, the pairs of a whole batch sat in a handful of cache lines and every fold of every block went through the same
// L2 channel's atomic unit (k_ln_ort: 26 us against 6 for the plain LayerNorm); one slot per utterance spreads them.
#define QV_MM_STRIDE 64

static __host__ __device__ __forceinline__ uint32_t fenc(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
static __host__ __device__ __forceinline__ float fdec(uint32_t e) {
    const uint32_t u = e ^ ((e >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __builtin_bit_cast(float, u);
}

struct QParam { float scale, zp, inv; };   // zp: integer-valued, 0..255; inv = 1 / scale (IEEE division, once)

// DynamicQuantizeLinear's parameters from one {min, max} key pair (float32 arithmetic, IEEE division)
static __device__ __forceinline__ QParam dql_param(const uint32_t *__restrict__ mm) {
    const float mn = fdec(mm[0]), mx = fdec(mm[1]);
    QParam p;
    p.scale = (mx - mn) / 255.0f;
    if (p.scale == 0.f) { p.scale = 1.0f; p.zp = 0.f; p.inv = 1.0f; return p; }
    const float z = rintf(-mn / p.scale);
    p.zp = fminf(fmaxf(z, 0.f), 255.f);
    p.inv = 1.0f / p.scale;
    return p;
}
// x -> x_q (0..255, as float): saturate(rne(x / scale) + zp), THE IEEE quotient's rounding, at a third of its cost.
// q = x * (1 / scale) is within 1.5 ulp of the true quotient (both factors correctly rounded); wherever the quotient lands
// it only matters which integer rne() picks, and q can pick another one than x / scale only if a rounding boundary
// k + 0.5 lies between them -- i.e. within 1.5 ulp(q) of q.  Those elements (2 in 1,000) take the division; the guard
// (1e-3 absolute + 1e-6 relative) is > 1.5 ulp for every |q| that survives the saturation and far beyond it.
// (the centred form x_q - zp = clamp(rne(x / scale), -zp, 255 - zp) is what the integer convolutions consume: exact
// integers either way, two operations fewer per value)
static __device__ __forceinline__ float quant_c(float x, const QParam &p) {
    const float q = x * p.inv;
    float t = rintf(q);
    if (fabsf(fabsf(q - t) - 0.5f) < 1e-3f + 1e-6f * fabsf(q)) t = rintf(x / p.scale);
    return __builtin_amdgcn_fmed3f(t, -p.zp, 255.f - p.zp);
}
static __device__ __forceinline__ float quant_u8(float x, const QParam &p) { return quant_c(x, p) + p.zp; }

// fold one value range into a site's pair; the plain read first drops most atomics (keys only ever move outwards)
static __device__ __forceinline__ void mm_fold(uint32_t *__restrict__ mm, float mn, float mx) {
    const uint32_t kn = fenc(mn), kx = fenc(mx);
    if (kn < __hip_atomic_load(mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(mm, kn);
    if (kx > __hip_atomic_load(mm + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(mm + 1, kx);
}

static __device__ __forceinline__ void mm_fold_keys(uint32_t *__restrict__ mm, uint32_t kn, uint32_t kx) {
    if (kn < __hip_atomic_load(mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(mm, kn);
    if (kx > __hip_atomic_load(mm + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(mm + 1, kx);
}

// ---- launchers (qv_ort.hip) ------------------------------------------------------------------------------------
// rows of a [rows][C] activation belong to utterances either through row_map (packed encoder rows: utterance << 16 |
// frame) or, with row_map == nullptr, densely: utterance = row / rows_per_utt, frame = (row % rows_per_utt) / f_per_t,
// valid iff frame < len[utterance].
struct RowOwner {
    const int32_t *row_map;
    const int32_t *len;
    int rows_per_utt, f_per_t;
};

void launch_mm_init(uint32_t *mm, size_t n_keys, hipStream_t s);
// f32 [rows][C] -> s8 [rows][C] with the owner's DynamicQuantizeLinear parameters (C % 16 == 0)
void launch_quant_rows(const float *x, int rows, int C, const RowOwner &own, const uint32_t *mm, int8_t *y, hipStream_t s);
// LayerNorm of packed rows: pass 1 folds the output range into mm (and keeps the f32 output when y32 != nullptr),
// pass 2 recomputes the same values and writes them quantised
void launch_ln_minmax(const float *x, const float *g, const float *b, int M, const int32_t *row_map, uint32_t *mm, float *y32,
                      hipStream_t s);
void launch_ln_quant(const float *x, const float *g, const float *b, int M, const int32_t *row_map, const uint32_t *mm, int8_t *y,
                     hipStream_t s);
// range of an f32 [M][512] activation of packed rows (the encoder output in front of the CTC head)
void launch_rows_minmax(const float *x, int M, const int32_t *row_map, uint32_t *mm, hipStream_t s);
// conv module: depthwise Conv1d(512, k 9) on quantised GLU output, + BatchNorm (eval) + Swish -> f32, range folded
void launch_dwconv1d_ort(const float *x, const float *wq /*[9][512] integer-valued*/, float w_scale, const float *bias,
                         const float *bn_alpha, const float *bn_beta, const int32_t *len, const int32_t *row_off,
                         const uint32_t *mm_in, uint32_t *mm_out, float *y, int t_max, int batch, hipStream_t s);
// front-end: per-utterance range of the normalised log-mel features
void launch_mel_minmax(const float *feats, const int32_t *n_samples, int tm_max, const double *stats, uint32_t *mm, int batch,
                       hipStream_t s);
// conv.0 (+ReLU) and conv.2 as integer convolutions.  pass 0: range of ReLU(conv.0) only; pass 1: the whole thing
void launch_sub01_ort(int pass, const float *feats, int tm_max, const int32_t *len_mel, const double *stats, const float *w0q,
                      float w0_scale, const float *b0, const int32_t *len1, const float *w1q, float w1_scale, const float *b1,
                      const int32_t *len2, const uint32_t *mm_mel, uint32_t *mm_c0, uint32_t *mm_c1, float *out, int t2_max,
                      int batch, hipStream_t s);
// conv.5: depthwise 3x3 / s2 on the quantised ReLU(conv.3) output -> f32, range folded
void launch_dwconv2d_ort(const float *in, int tin_max, int fin, const int32_t *len_in, const float *wq, float w_scale,
                         const float *bias, const int32_t *len_out, const uint32_t *mm_in, uint32_t *mm_out, float *out,
                         int tout_max, int fout, int batch, hipStream_t s);
