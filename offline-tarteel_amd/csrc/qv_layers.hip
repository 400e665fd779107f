// qv_layers.hip -- non-GEMM kernels of the FastConformer forward: log-mel front-end, strided
// conv subsampling, LayerNorm, rel-pos attention, depthwise conv, log-softmax.
//
// All of them are HBM/LDS-bound elementwise or small-reduction kernels except the attention,
// which runs QK^T, the rel-pos term and PV on v_mfma_f32_32x32x16_f16 with operands read
// straight from HBM/L2 in fragment order (K-contiguous 16-B loads; V is produced already
// transposed by the QKV GEMM epilogue so that PV's B operand is K-contiguous too).

#include "qv_layers.h"
#include <type_traits>

#include <atomic>
#include "qv_dev_util.h"
#include "qv_logmel_reg.h"

#include <math.h>

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ------------------------------------------------------------------ front-end ----------
// One wave per STFT frame: pre-emphasis + reflect padding + Hann window on load; the 512 real
// samples are packed as 256 complex ones (z[n] = x[2n] + i x[2n+1]), transformed with a 256-point
// radix-2 Stockham FFT in LDS (8 passes, 2 butterflies per lane per pass) and unpacked to the 257
// bins of the real transform; power spectrum, sparse mel projection (each filter touches <= 32
// bins), log(x + 2^-24).
__global__ __launch_bounds__(256) void k_logmel(const float *__restrict__ audio, int64_t n_max,
                                                const int32_t *__restrict__ n_samples, const FrontendTab ft,
                                                float *__restrict__ feats, int tm_max) {
    __shared__ float2 buf[4][2][256];
    __shared__ float pw[4][264];
    __shared__ float2 tw[256];   // the twiddle table: every FFT pass fetched its factors from global memory (waves parked 79 %)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.y, t = blockIdx.x * 4 + wave;
    const int n = n_samples[b];
    const int tm = n / 160 + 1;
    tw[threadIdx.x] = ft.twiddle[threadIdx.x];
    __syncthreads();
    if (t >= tm) return;  // frames past the utterance are zeroed by the normalisation kernel
    const float *x = audio + (size_t)b * n_max;
    float2 *A = buf[wave][0], *Bf = buf[wave][1];
    auto sample = [&](int i) {
        int s = t * 160 - 256 + i;
        if (s < 0) s = -s;
        if (s >= n) s = 2 * (n - 1) - s;
        s = s < 0 ? 0 : s;
        float y = x[s] - (s > 0 ? 0.97f * x[s - 1] : 0.f);
        return y * ft.window[i];
    };
    for (int i = lane; i < 256; i += 64) A[i] = make_float2(sample(2 * i), sample(2 * i + 1));
    // Stockham autosort, radix 2, N = 256: pass p (len = 1 << p): out[j*2*len + k] , out[... + len]
    float2 *src = A, *dst = Bf;
    for (int p = 0; p < 8; ++p) {
        int len = 1 << p;  // half-size of the butterflies produced so far
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < 128; i += 64) {
            int k = i & (len - 1), j = i >> p;  // j: group, k: index within group
            float2 u = src[j * len + k], v = src[j * len + k + 128];
            // twiddle w = exp(-2 pi i * k / (2 len)) from the 512-point table
            float2 w = tw[k * (256 >> p)];
            float2 vw = make_float2(__builtin_fmaf(v.x, w.x, -(v.y * w.y)), __builtin_fmaf(v.x, w.y, v.y * w.x));
            dst[j * 2 * len + k] = make_float2(u.x + vw.x, u.y + vw.y);
            dst[j * 2 * len + k + len] = make_float2(u.x - vw.x, u.y - vw.y);
        }
        float2 *tmp = src; src = dst; dst = tmp;
    }
    __builtin_amdgcn_wave_barrier();
    // X[k] = E[k] - i W^k O[k],  E = (Z[k] + conj Z[256-k]) / 2,  O = (Z[k] - conj Z[256-k]) / 2,  W = exp(-2 pi i / 512)
    for (int k = lane; k < 257; k += 64) {
        float2 X;
        if (k == 0 || k == 256) {
            float2 z0 = src[0];
            X = make_float2(k == 0 ? z0.x + z0.y : z0.x - z0.y, 0.f);
        } else {
            float2 zk = src[k], zc = src[256 - k];
            float2 E = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
            float2 O = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y + zc.y));
            float2 w = tw[k];
            float2 P = make_float2(__builtin_fmaf(w.x, O.x, -(w.y * O.y)), __builtin_fmaf(w.x, O.y, w.y * O.x));
            X = make_float2(E.x + P.y, E.y - P.x);
        }
        float mag = sqrtf(X.x * X.x + X.y * X.y);
        pw[wave][k] = mag * mag;
    }
    __builtin_amdgcn_wave_barrier();
    float *out = feats + ((size_t)b * tm_max + t) * QV_NMEL;
    for (int m = lane; m < QV_NMEL; m += 64) {
        int lo = ft.mel_lo[m], cnt = ft.mel_cnt[m];
        const float *w = ft.mel_w + m;        // tap-major [32][80]
        float acc = 0.f;
        for (int k = 0; k < cnt; ++k) acc += w[k * QV_NMEL] * pw[wave][lo + k];
        out[m] = logf(acc + 5.9604644775390625e-08f);
    }
}

// per-feature sum / sum of squares over the valid frames (f64 accumulation); the normalisation itself is applied by conv0
// while it loads its input rows, so the features make no extra round trip through HBM.
// Each of the MS_CHUNKS blocks of an utterance writes its partial sums and k_melstats_sum adds them in chunk order.  (Until
// round 4 the blocks added their partials to the result with f64 atomics, i.e. in whatever order they finished: the last
// bit of a sum then depends on what else the GPU is running.  This was the SECOND, independent source of run-to-run
// differences found while chasing the 38-of-1,500 soak mismatches of round 4; the first and larger one was the packed-FP32
// hazard in k_logmel's unpack step that round 5 pinned down -- build.py, tests/test_gpu_interference.py.)
#define MS_CHUNKS 16
__global__ __launch_bounds__(320) void k_melstats(const float *__restrict__ feats, const int32_t *__restrict__ n_samples,
                                                  int tm_max, double *__restrict__ part /*[B][MS_CHUNKS][80][2]*/) {
    __shared__ double p1[4][QV_NMEL], p2[4][QV_NMEL];
    const int b = blockIdx.y, f = threadIdx.x % QV_NMEL, g = threadIdx.x / QV_NMEL;  // 4 time groups
    const int tm = n_samples[b] / 160 + 1;
    const int per = (tm + MS_CHUNKS - 1) / MS_CHUNKS, t0 = blockIdx.x * per, t1 = min(tm, t0 + per);
    const float *x = feats + (size_t)b * tm_max * QV_NMEL;
    double s1 = 0.0, s2 = 0.0;
    for (int t = t0 + g; t < t1; t += 4) { double v = x[t * QV_NMEL + f]; s1 += v; s2 += v * v; }
    p1[g][f] = s1;
    p2[g][f] = s2;
    __syncthreads();
    if (g == 0) {     // an empty chunk writes zeros
        double *o = part + (((size_t)b * MS_CHUNKS + blockIdx.x) * QV_NMEL + f) * 2;
        o[0] = p1[0][f] + p1[1][f] + p1[2][f] + p1[3][f];
        o[1] = p2[0][f] + p2[1][f] + p2[2][f] + p2[3][f];
    }
}
__global__ __launch_bounds__(2 * QV_NMEL) void k_melstats_sum(const double *__restrict__ part, double *__restrict__ acc /*[B][80][2]*/) {
    const int b = blockIdx.x, i = threadIdx.x;      // i = feature * 2 + {sum, sum of squares}
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < MS_CHUNKS; ++c) s += part[((size_t)b * MS_CHUNKS + c) * (2 * QV_NMEL) + i];
    acc[(size_t)b * 2 * QV_NMEL + i] = s;
}

// materialise the normalised features (only for the debug tap / parity tests)
__global__ __launch_bounds__(320) void k_melapply(const float *__restrict__ feats, const int32_t *__restrict__ n_samples,
                                                  int tm_max, const double *__restrict__ acc, float *__restrict__ out) {
    const int b = blockIdx.y, f = threadIdx.x % QV_NMEL, g = threadIdx.x / QV_NMEL;
    const int tm = n_samples[b] / 160 + 1;
    float mean, rstd;
    mel_mean_rstd(acc, b, f, tm, mean, rstd);
    for (int t = blockIdx.x * 4 + g; t < tm_max; t += gridDim.x * 4) {
        size_t i = ((size_t)b * tm_max + t) * QV_NMEL + f;
        out[i] = t < tm ? (feats[i] - mean) * rstd : 0.f;
    }
}

// ------------------------------------------------------------------ subsampling --------
// conv0: Conv2d(1->256, 3x3, s2, p1) + ReLU on [B][Tm][80] -> channels-last f16 [B][T1][40][256].
// Round 4: on the f32 matrix pipe.  As 9 FMAs per output on the VALU this layer was 6 GFLOP per batch at 17 % of the
// packed-FMA peak, issue-bound (k_sub01: 36 packed FMAs + ~30 address / LDS instructions per position and thread).
// It is a [channels] x [9 taps] x [positions] product: v_mfma_f32_32x32x2_f32 takes two taps per instruction, so a
// 32-channel x 32-position tile is five of them (the tenth tap is zero) with the bias as the initial accumulator --
// float32 operands and accumulation like before, only the summation order inside the instruction is the hardware's.
// A operand: lane (l31 = channel, hi = tap parity) holds w[2j + hi][channel]; B operand: lane (l31 = position, hi)
// holds the input sample of tap 2j + hi at that position; D register r of lane (l31 = position, hi) is channel
// (r & 3) + 8 (r >> 2) + 4 hi.  conv0_taps / conv0_tile are shared by the fused kernel and the two-kernel cross-check
// path, which therefore still agree bit for bit.
__device__ __forceinline__ void conv0_weights(const float *__restrict__ wt /*[9][256]*/, const float *__restrict__ bias, int ch0,
                                              int l31, int hi, float wa[5], float bc[16]) {
#pragma unroll
    for (int j = 0; j < 5; ++j) wa[j] = 2 * j + hi < 9 ? wt[(2 * j + hi) * QV_SUBC + ch0 + l31] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) bc[r] = bias[ch0 + (r & 3) + 8 * (r >> 2) + 4 * hi];
}
// (conv0_tap_offsets / conv0_taps, the B operand fetch, live in qv_dev_util.h: precision 2's front end uses them too)
__device__ __forceinline__ f32x16 conv0_tile(const float wa[5], const float bc[16], const float xb[5]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bc[r];
#pragma unroll
    for (int j = 0; j < 5; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j], xb[j], acc, 0, 0, 0);
    return acc;
}

// ReLU + conversion of a D tile's 16 registers: convert first (round to nearest even, v_cvt_pk_f16_f32), then clamp the
// f16 BITS as signed integers (a negative float is a negative int16; one v_pk_max_i16 per two channels).
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half4 conv0_relu4(const f32x16 &acc, int q) {
    const float4_t v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    const half4 h = __builtin_convertvector(v, half4);
    short4_t b = __builtin_bit_cast(short4_t, h);
    b = __builtin_elementwise_max(b, (short4_t){0, 0, 0, 0});
    return __builtin_bit_cast(half4, b);
}

// two-kernel path (QVERSE_SUB_UNFUSED=1, the cross-check of k_sub01): block = one output row t1; the 3 input rows sit in
// LDS; wave w owns channel tiles 2w and 2w + 1, both position tiles (40 positions = 32 + 8).
__global__ __launch_bounds__(256) void k_conv0(const float *__restrict__ feats, int tm_max, const int32_t *__restrict__ len_in,
                                               const double *__restrict__ stats, const float *__restrict__ wt /*[9][256]*/,
                                               const float *__restrict__ bias, half_t *__restrict__ out, int t1_max) {
    __shared__ float rows[3][QV_NMEL + 2];
    __shared__ float mean_s[QV_NMEL], rstd_s[QV_NMEL];
    const int b = blockIdx.z, t1 = blockIdx.y, tid = threadIdx.x;
    const int tin = len_in[b];
    const float *x = feats + (size_t)b * tm_max * QV_NMEL;
    if (tid < QV_NMEL) mel_mean_rstd(stats, b, tid, tin, mean_s[tid], rstd_s[tid]);
    __syncthreads();
    // per-feature normalisation (NeMo normalize_batch "per_feature") applied on load; frames past
    // the utterance and the conv padding read as 0
    for (int i = tid; i < 3 * (QV_NMEL + 2); i += 256) {
        int dt = i / (QV_NMEL + 2), f = i % (QV_NMEL + 2) - 1, t = 2 * t1 - 1 + dt;
        rows[dt][f + 1] = (t >= 0 && t < tin && f >= 0 && f < QV_NMEL) ? (x[t * QV_NMEL + f] - mean_s[f]) * rstd_s[f] : 0.f;
    }
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    int koff[5];
    conv0_tap_offsets(hi, koff);
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int ch0 = (2 * wave + ct) * 32;
        float wa[5], bc[16];
        conv0_weights(wt, bias, ch0, l31, hi, wa, bc);
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int f1 = pt * 32 + l31, f1c = f1 < 40 ? f1 : 39;
            float xb[5];
            conv0_taps(&rows[0][0], 0, f1c, koff, xb);
            const f32x16 acc = conv0_tile(wa, bc, xb);
            if (f1 < 40) {
                half_t *o = out + (((size_t)b * t1_max + t1) * 40 + f1) * QV_SUBC + ch0 + 4 * hi;
#pragma unroll
                for (int q = 0; q < 4; ++q) *(half4 *)(o + 8 * q) = conv0_relu4(acc, q);
            }
        }
    }
}

// depthwise Conv2d(256, 3x3, s2, p1, groups=256) on channels-last f16; rows t >= len_in[b] read as 0.
// Same ownership as conv0: 8 channels per thread, weights [9][256] in registers.
#define DW2_TT 4
__global__ __launch_bounds__(256) void k_dwconv2d(const half_t *__restrict__ in, int tin_max, int fin,
                                                  const int32_t *__restrict__ len_in, const float *__restrict__ wt,
                                                  const float *__restrict__ bias, half_t *__restrict__ out, int tout_max,
                                                  int fout) {
    // a block owns DW2_TT consecutive output frames: the 80 weights a thread fetches serve DW2_TT x fout / 8 outputs
    // instead of fout / 8 (they were most of the block's L2 traffic), and with fout = 10 the 8 position slots of a
    // channel group are all busy (40 positions / 8) instead of 10 of 16
    const int b = blockIdx.z, to0 = blockIdx.y * DW2_TT, tid = threadIdx.x;
    const int tin = len_in[b];
    const int c0 = (tid & 31) * 8, fl = tid >> 5;
    float w[9][8], bs[8];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        f32x4 w0 = *(const f32x4 *)(wt + k * QV_SUBC + c0), w1 = *(const f32x4 *)(wt + k * QV_SUBC + c0 + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { w[k][c] = w0[c]; w[k][4 + c] = w1[c]; }
    }
    {
        f32x4 b0 = *(const f32x4 *)(bias + c0), b1 = *(const f32x4 *)(bias + c0 + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { bs[c] = b0[c]; bs[4 + c] = b1[c]; }
    }
    for (int p = fl; p < DW2_TT * fout; p += 8) {
        const int to = to0 + p / fout, fo = p % fout;
        if (to >= tout_max) break;
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = bs[c];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            int t = 2 * to - 1 + dt;
            if (t < 0 || t >= tin) continue;
#pragma unroll
            for (int df = 0; df < 3; ++df) {
                int f = 2 * fo - 1 + df;
                if (f < 0 || f >= fin) continue;
                half8 v = *(const half8 *)(in + (((size_t)b * tin_max + t) * fin + f) * QV_SUBC + c0);
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = __builtin_fmaf(w[dt * 3 + df][c], (float)v[c], acc[c]);
            }
        }
        half8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = (half_t)acc[c];
        *(half8 *)(out + (((size_t)b * tout_max + to) * fout + fo) * QV_SUBC + c0) = o;
    }
}

// zero rows t >= len[b] of a channels-last f16 activation (so the next strided stage and the
// out-projection see exactly what the unpadded single-utterance run sees)
// conv0 + ReLU + first depthwise conv in one kernel: the [B][T1][40][256] activation (657 MB at
// B = 64 x 10 s) never goes to HBM.  Block = (64-channel group, 4 output frames, utterance): the
// 19 normalised mel rows it needs sit in LDS, the 9 x 40 x 64 conv0 tile is computed once into LDS
// as f16 (same rounding as the two-kernel path), then the depthwise 3x3/s2 reads it from there.
// Both convolutions are per-channel, so the only redundancy is the one-row halo (9 rows per 8).
#define SUB_TT 4                    // c1 frames per block
#define SUB_R1 (2 * SUB_TT + 1)     // c0 rows per block
#define SUB_RM (2 * SUB_R1 + 1)     // mel rows per block
#define SUB_CG 64                   // channels per block
__global__ __launch_bounds__(256) void k_sub01(const float *__restrict__ feats, int tm_max, const int32_t *__restrict__ len_mel,
                                               const double *__restrict__ stats, const float *__restrict__ w0t,
                                               const float *__restrict__ b0, const int32_t *__restrict__ len1,
                                               const float *__restrict__ w1t, const float *__restrict__ b1,
                                               half_t *__restrict__ out, int t2_max) {
    __shared__ __attribute__((aligned(16))) float rows[SUB_RM][QV_NMEL + 2];
    __shared__ float mean_s[QV_NMEL], rstd_s[QV_NMEL];
    // conv0 tile, position-major: position p = row * 40 + f owns 64 channels = eight 16-byte chunks; chunk c is stored at
    // ((c + p) & 7) so that the 8-byte stores of the conv0 epilogue (32 positions x 2 halves per wave) spread over the banks
    __shared__ __attribute__((aligned(16))) half_t tile[SUB_R1 * 40][SUB_CG];
    // both convolutions' taps (rows 0 .. 8) and biases (row 9) for all 256 channels: fetched once per block, next to the mel
    // rows' latency (per channel group from global memory they were 72 load instructions per thread and a latency in
    // front of every group's matrix products)
    __shared__ __attribute__((aligned(16))) float w0s[10][QV_SUBC], w1s[10][QV_SUBC];
    // ~73 KB of static LDS: gfx950 only (160 KB per CU, two blocks resident); every other target stops at 64 KB per block
    static_assert(sizeof(rows) + sizeof(tile) + sizeof(w0s) + sizeof(w1s) + sizeof(mean_s) + sizeof(rstd_s) <= 80 * 1024,
                  "k_sub01: two blocks per CU need <= 80 KB of LDS each");
    const int b = blockIdx.z, t2_0 = blockIdx.y * SUB_TT, tid = threadIdx.x;
    const int tin = len_mel[b], l1 = len1[b];
    const float *x = feats + (size_t)b * tm_max * QV_NMEL;
    if (tid < QV_NMEL) mel_mean_rstd(stats, b, tid, tin, mean_s[tid], rstd_s[tid]);
    for (int i = tid; i < 10 * QV_SUBC / 4; i += 256) {
        const int k = i / (QV_SUBC / 4), c = (i % (QV_SUBC / 4)) * 4;
        *(f32x4 *)&w0s[k][c] = *(const f32x4 *)((k < 9 ? w0t + k * QV_SUBC : b0) + c);
        *(f32x4 *)&w1s[k][c] = *(const f32x4 *)((k < 9 ? w1t + k * QV_SUBC : b1) + c);
    }
    __syncthreads();
    const int t1_0 = 2 * t2_0 - 1;         // first c0 row of the tile
    const int tm_0 = 2 * t1_0 - 1;         // first mel row
    for (int i = tid; i < SUB_RM * (QV_NMEL + 2); i += 256) {
        int r = i / (QV_NMEL + 2), f = i % (QV_NMEL + 2) - 1, t = tm_0 + r;
        rows[r][f + 1] = (t >= 0 && t < tin && f >= 0 && f < QV_NMEL) ? (x[t * QV_NMEL + f] - mean_s[f]) * rstd_s[f] : 0.f;
    }
    const int c8 = (tid & 7) * 8, pl = tid >> 3;   // depthwise stage: 8 channels per thread, 32 positions per pass
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    int koff[5];
    conv0_tap_offsets(hi, koff);
    // Round 4: the block walks the four 64-channel groups itself (the grid had one block per group: four blocks each
    // waited for the statistics, fetched and normalised the same 19 mel rows -- three global latencies in front of
    // ~7 k cycles of arithmetic, the larger part of a block's life).  The tile is reused from group to group.
    for (int cg = 0; cg < QV_SUBC; cg += SUB_CG) {
        // ---- conv0 + ReLU into the LDS tile on the f32 matrix pipe (rows outside [0, l1) are the depthwise conv's zero
        // padding): the 360 positions are 12 tiles of 32 (the last one 8 wide), wave w owns tiles w, w + 4, w + 8 and
        // both 32-channel halves of the group
        {
            __syncthreads();        // the mel rows and the weights are in LDS / the previous group's depthwise stage has left the tile
            float wa[2][5], bc[2][16];
            conv0_weights(&w0s[0][0], &w0s[9][0], cg, l31, hi, wa[0], bc[0]);
            conv0_weights(&w0s[0][0], &w0s[9][0], cg + 32, l31, hi, wa[1], bc[1]);
#pragma unroll
            for (int pt = wave; pt < (SUB_R1 * 40 + 31) / 32; pt += 4) {
                const int p = pt * 32 + l31, pc = p < SUB_R1 * 40 ? p : SUB_R1 * 40 - 1;
                const int r = pc / 40, f1 = pc - r * 40;
                float xb[5];
                conv0_taps(&rows[0][0], 2 * r, f1, koff, xb);
                half_t *trow = &tile[pc][4 * hi];       // lanes past the last position rewrite position 359 with its own values
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const f32x16 acc = conv0_tile(wa[ct], bc[ct], xb);
#pragma unroll
                    for (int q = 0; q < 4; ++q) *(half4 *)(trow + (((ct * 4 + q + pc) & 7) << 3)) = conv0_relu4(acc, q);
                }
            }
        }
        // rows outside [0, l1) are the depthwise conv's zero padding, not conv0 outputs: only the first block of an
        // utterance and the ones at its end have any, and they zero them behind the products
        if (t1_0 < 0 || t1_0 + SUB_R1 > l1) {
            __syncthreads();
            for (int i = tid; i < SUB_R1 * 40 * (SUB_CG / 8); i += 256) {
                const int pp = i >> 3, t1 = t1_0 + pp / 40;
                if (t1 < 0 || t1 >= l1) *(f32x4 *)&tile[pp][(i & 7) << 3] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();            // conv0 done: the tile is complete
        // ---- depthwise 3x3 stride 2 over the tile
        float w[9][8], bs[8];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const f32x4 wa = *(const f32x4 *)&w1s[k][cg + c8], wb = *(const f32x4 *)&w1s[k][cg + c8 + 4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { w[k][c] = wa[c]; w[k][4 + c] = wb[c]; }
        }
        {
            const f32x4 ba = *(const f32x4 *)&w1s[9][cg + c8], bb = *(const f32x4 *)&w1s[9][cg + c8 + 4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { bs[c] = ba[c]; bs[4 + c] = bb[c]; }
        }
        for (int p = pl; p < SUB_TT * 20; p += 32) {
            int tl = p / 20, fo = p - tl * 20, t2 = t2_0 + tl;
            if (t2 >= t2_max) continue;
            float acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = bs[c];
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int df = 0; df < 3; ++df) {
                    int f = 2 * fo - 1 + df;
                    if (f < 0 || f >= 40) continue;
                    const int pp = (2 * tl + dt) * 40 + f;
                    half8 v = *(const half8 *)&tile[pp][(((tid & 7) + pp) & 7) << 3];
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c] = __builtin_fmaf(w[dt * 3 + df][c], (float)v[c], acc[c]);
                }
            half8 o;
#pragma unroll
            for (int c = 0; c < 8; ++c) o[c] = (half_t)acc[c];
            *(half8 *)(out + (((size_t)b * t2_max + t2) * 20 + fo) * QV_SUBC + cg + c8) = o;
        }
    }
}

// dense [B][t_max][row_elems] -> packed [sum of len][row_elems]: utterance b's valid frames become
// rows row_off[b] .. row_off[b] + len[b] - 1 (from here on padding frames do not exist).  Also
// records each packed row's owner, row_map[row] = utterance << 16 | frame, for the kernels that
// have to go back from a row to its utterance (V transpose in the QKV epilogue, log-softmax).
__global__ void k_pack_rows(const half_t *__restrict__ x, int t_max, int row_elems, const int32_t *__restrict__ len,
                            const int32_t *__restrict__ row_off, half_t *__restrict__ y, int32_t *__restrict__ row_map) {
    const int b = blockIdx.z, t = blockIdx.y;
    if (t >= len[b]) return;
    const half_t *p = x + ((size_t)b * t_max + t) * row_elems;
    half_t *q = y + ((size_t)row_off[b] + t) * row_elems;
    if (threadIdx.x == 0) row_map[row_off[b] + t] = (b << 16) | t;
    for (int i = threadIdx.x * 8; i < row_elems; i += blockDim.x * 8) *(half8 *)(q + i) = *(const half8 *)(p + i);
}

// ------------------------------------------------------------------ LayerNorm ----------
// one wave per row of 512: f32 in -> f16 out (GEMM operand).  Two-pass variance in registers.
__device__ __forceinline__ void ln_row(const float v[8], const float *__restrict__ gam, const float *__restrict__ bet, int lane,
                                       float o[8]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    float mu = wave_sum(s) * (1.f / QV_D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { float d = v[i] - mu; q += d * d; }
    float rs = rsqrtf(wave_sum(q) * (1.f / QV_D) + 1e-5f);
    f32x4 g0 = *(const f32x4 *)(gam + lane * 8), g1 = *(const f32x4 *)(gam + lane * 8 + 4);
    f32x4 b0 = *(const f32x4 *)(bet + lane * 8), b1 = *(const f32x4 *)(bet + lane * 8 + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = (v[i] - mu) * rs * g0[i] + b0[i]; o[4 + i] = (v[4 + i] - mu) * rs * g1[i] + b1[i]; }
}

// One wave normalises LN_ROWS consecutive rows: all their loads (and the parameters) are in flight
// before the first reduction, and a wave lives for LN_ROWS rows instead of one.
#ifndef LN_ROWS
#define LN_ROWS 2
#endif
__global__ __launch_bounds__(256) void k_layernorm(const float *__restrict__ x, const float *__restrict__ gam,
                                                   const float *__restrict__ bet, half_t *__restrict__ y, int M) {
    const int row0 = (xcd_order(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6)) * LN_ROWS, lane = threadIdx.x & 63;
    if (row0 >= M) return;
    f32x4 a[LN_ROWS], c[LN_ROWS];
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        const int row = row0 + r < M ? row0 + r : M - 1;
        const float *p = x + (size_t)row * QV_D + lane * 8;
        a[r] = *(const f32x4 *)p; c[r] = *(const f32x4 *)(p + 4);
    }
    const LnParam pr = ln_param(gam, bet, lane);
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        if (row0 + r >= M) break;
        float v[8] = {a[r][0], a[r][1], a[r][2], a[r][3], c[r][0], c[r][1], c[r][2], c[r][3]}, o[8];
        ln_row_p(v, pr, o);
        half8 h;
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = (half_t)o[i];
        *(half8 *)(y + (size_t)(row0 + r) * QV_D + lane * 8) = h;
    }
}

// x <- LN_out(x) (f32, in place: the next layer's residual stream), y <- LN_next(x) (f16); with g2 == nullptr and y set
// (last layer) y is the f16 copy of LN_out(x) itself: the CTC head's operand, no separate conversion pass
__global__ __launch_bounds__(256) void k_layernorm2(float *__restrict__ x, const float *__restrict__ g1, const float *__restrict__ b1,
                                                    const float *__restrict__ g2, const float *__restrict__ b2,
                                                    half_t *__restrict__ y, int M) {
    const int row0 = (xcd_order(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6)) * LN_ROWS, lane = threadIdx.x & 63;
    if (row0 >= M) return;
    f32x4 a[LN_ROWS], c[LN_ROWS];
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        const int row = row0 + r < M ? row0 + r : M - 1;
        const float *p = x + (size_t)row * QV_D + lane * 8;
        a[r] = *(const f32x4 *)p; c[r] = *(const f32x4 *)(p + 4);
    }
    const LnParam p1 = ln_param(g1, b1, lane);
    LnParam p2 = p1;
    const bool second = y && g2;
    if (second) p2 = ln_param(g2, b2, lane);
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        if (row0 + r >= M) break;
        float *p = x + (size_t)(row0 + r) * QV_D + lane * 8;
        float v[8] = {a[r][0], a[r][1], a[r][2], a[r][3], c[r][0], c[r][1], c[r][2], c[r][3]}, o[8], o2[8];
        ln_row_p(v, p1, o);
        *(f32x4 *)p = f32x4{o[0], o[1], o[2], o[3]};
        *(f32x4 *)(p + 4) = f32x4{o[4], o[5], o[6], o[7]};
        if (y) {
            if (second) ln_row_p(o, p2, o2);
            else {
#pragma unroll
                for (int i = 0; i < 8; ++i) o2[i] = o[i];
            }
            half8 h;
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] = (half_t)o2[i];
            *(half8 *)(y + (size_t)(row0 + r) * QV_D + lane * 8) = h;
        }
    }
}

// f32 -> f16 copy of the final encoder output (CTC head operand)
__global__ void k_to_half(const float *__restrict__ x, half_t *__restrict__ y, size_t n8) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    f32x4 a = *(const f32x4 *)(x + i * 8), c = *(const f32x4 *)(x + i * 8 + 4);
    half8 h = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)c[0], (half_t)c[1], (half_t)c[2], (half_t)c[3]};
    *(half8 *)(y + i * 8) = h;
}

// f16 -> f32 (debug taps)
__global__ void k_to_float(const half_t *__restrict__ x, float *__restrict__ y, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (float)x[i];
}

// ------------------------------------------------------------------ attention ----------
// RelPositionMultiHeadAttention core for one (utterance, head): flash-style online softmax over
// 32-key tiles, one wave per 32-query tile, everything on v_mfma_f32_32x32x16_f16.
//   AC[i,j]  = (q_i + u) . k_j
//   BD[i,j]  = (q_i + v) . p_{T-1-i+j}          (rel_shift of the [T, 2T-1] product)
//   out      = softmax((AC + BD) / 8, keys < len) V
// All products are computed TRANSPOSED (keys / positions / head-dim on the MFMA row axis, queries
// on the column axis), so a lane owns ONE query: its 16 accumulator registers are 16 keys of that
// query, the softmax reductions are in-register plus one cross-half shuffle, the running max / sum
// are per-lane scalars, and exp(S^T) is already in the B-operand layout of the P.V product (the
// contraction index is permuted identically on the V side, which only changes which 8-byte pieces
// of V^T a lane loads).  The rel-pos term needs raw[c = 31 - i + j]: the two raw tiles go through
// a per-wave LDS slab (row = query, odd stride) and come back skewed.
#define ATT_LDS_LD 67  // floats per query row of the skew slab (odd: conflict-free skewed reads)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_attention(const half_t *__restrict__ qk, const half_t *__restrict__ vt,
                                                   const half_t *__restrict__ pos, int pos_ld, const float *__restrict__ bias_u,
                                                   const float *__restrict__ bias_v, const int32_t *__restrict__ len,
                                                   const int32_t *__restrict__ row_off, half_t *__restrict__ out, int t_max,
                                                   int t_pad) {
    __shared__ float slab[4][32 * ATT_LDS_LD];
    const int b = blockIdx.y, h = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int T = len[b];
    const size_t row0 = (size_t)row_off[b];   // activations are packed: utterance b owns rows [row0, row0 + T)
    const int l31 = lane & 31, hi = lane >> 5;
    const half_t *qb = qk + row0 * (2 * QV_D) + h * QV_DK;                           // q row stride 1024
    const half_t *kb = qb + QV_D;
    const half_t *vb = vt + ((size_t)b * QV_D + h * QV_DK) * t_pad;                  // [64][t_pad]
    const half_t *pb = pos + h * QV_DK;                                             // [2*t_max-1][pos_ld]
    const int n_qt = (T + 31) >> 5, n_kt = (T + 31) >> 5;
    float *sl = slab[wave];
    for (int qt = wave; qt < n_qt; qt += 4) {
        const int i0 = qt * 32;
        half_t *orow = out + (row0 + i0) * QV_D + h * QV_DK;
        // B fragments (queries on the column axis): (q+u) and (q+v), row i0 + l31, d = ks*16 + hi*8..
        half8 qu[4], qv[4];
        {
            int qi = i0 + l31;
            qi = qi < T ? qi : T - 1;   // never touch another utterance's rows
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                int d = ks * 16 + hi * 8;
                half8 q8 = *(const half8 *)(qb + (size_t)qi * (2 * QV_D) + d);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float qf = (float)q8[e];
                    qu[ks][e] = (half_t)(qf + bias_u[h * QV_DK + d + e]);
                    qv[ks][e] = (half_t)(qf + bias_v[h * QV_DK + d + e]);
                }
            }
        }
        f32x16 o0, o1;  // O^T: rows d (0..31 / 32..63), column = this lane's query
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
        float m_run = -1e30f, l_run = 0.f;
        // fragments of one key tile: K rows, the two position tiles, V^T pieces (in the key order of the P.V operand)
        struct TileKP { half8 kf[4], p0[4], p1[4]; };
        struct TileV { half4 a0[2], a1[2], c0[2], c1[2]; };
        auto fetch_kp = [&](int kt, TileKP &f) {
            const int j0 = kt * 32;
            int kj = j0 + l31;
            kj = kj < T ? kj : T - 1;
            // relative-position rows for this tile: rr0 + c, c in [0,64)
            const int rr0 = t_max - 1 - i0 - 31 + j0;
            int pr0 = rr0 + l31, pr1 = rr0 + 32 + l31;
            pr0 = pr0 < 0 ? 0 : (pr0 > 2 * t_max - 2 ? 2 * t_max - 2 : pr0);
            pr1 = pr1 < 0 ? 0 : (pr1 > 2 * t_max - 2 ? 2 * t_max - 2 : pr1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                int d = ks * 16 + hi * 8;
                f.kf[ks] = *(const half8 *)(kb + (size_t)kj * (2 * QV_D) + d);
                f.p0[ks] = *(const half8 *)(pb + (size_t)pr0 * pos_ld + d);
                f.p1[ks] = *(const half8 *)(pb + (size_t)pr1 * pos_ld + d);
            }
        };
        auto fetch_v = [&](int kt, TileV &f) {
            const int j0 = kt * 32;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                int jb = j0 + 16 * ks + 4 * hi;  // < t_pad
                f.a0[ks] = *(const half4 *)(vb + (size_t)l31 * t_pad + jb);
                f.a1[ks] = *(const half4 *)(vb + (size_t)l31 * t_pad + jb + 8);
                f.c0[ks] = *(const half4 *)(vb + (size_t)(32 + l31) * t_pad + jb);
                f.c1[ks] = *(const half4 *)(vb + (size_t)(32 + l31) * t_pad + jb + 8);
            }
        };
        auto tile = [&](int kt, const TileKP &cur, const TileV &cv) {
            const int j0 = kt * 32;
            f32x16 st, r0, r1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; r0[r] = 0.f; r1[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.kf[ks], qu[ks], st, 0, 0, 0);   // S^T[jj][ii]
                r0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.p0[ks], qv[ks], r0, 0, 0, 0);   // raw^T[c][ii], c < 32
                r1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.p1[ks], qv[ks], r1, 0, 0, 0);   // raw^T[32 + c][ii]
            }
            // skew through LDS: slab[ii][c] <- raw^T[c][ii]; BD^T[jj][ii] = slab[ii][31 - ii + jj]
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int c = (r & 3) + 8 * (r >> 2) + 4 * hi;
                sl[l31 * ATT_LDS_LD + c] = r0[r];
                sl[l31 * ATT_LDS_LD + 32 + c] = r1[r];
            }
            __builtin_amdgcn_wave_barrier();
            float p[16];
            float mx = -1e30f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int jj = (r & 3) + 8 * (r >> 2) + 4 * hi;
                float bd = sl[l31 * ATT_LDS_LD + 31 - l31 + jj];
                float sc = (st[r] + bd) * 0.125f;
                p[r] = (j0 + jj < T) ? sc : -1e30f;
                mx = fmaxf(mx, p[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float m_new = fmaxf(m_run, mx);
            float corr = __expf(m_run - m_new);
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int jj = (r & 3) + 8 * (r >> 2) + 4 * hi;
                float e = (j0 + jj < T) ? __expf(p[r] - m_new) : 0.f;
                p[r] = e;
                sum += e;
            }
            sum += __shfl_xor(sum, 32);
            l_run = l_run * corr + sum;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }
            // P.V: B operand = exp(S^T) registers 8ks..8ks+7 (keys 16ks + 4hi + {0..3, 8..11});
            // A operand = V^T rows d with the same key order
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 pbf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pbf[e] = (half_t)p[8 * ks + e];
                const half4 a0 = cv.a0[ks], a1 = cv.a1[ks], c0 = cv.c0[ks], c1 = cv.c1[ks];
                half8 va = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                half8 vc = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pbf, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vc, pbf, o1, 0, 0, 0);
            }
        };
        for (int kt = 0; kt < n_kt; ++kt) {
            TileKP cur;
            TileV cv;
            fetch_kp(kt, cur);
            fetch_v(kt, cv);
            tile(kt, cur, cv);
        }
        // O^T column (query i0 + l31): d = (r&3) + 8*(r>>2) + 4*hi (+32)
        if (i0 + l31 < T) {
            float inv = 1.f / l_run;
            half_t *o = orow + (size_t)l31 * QV_D;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int d = 8 * q + 4 * hi;
                half4 h0, h1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { h0[e] = (half_t)(o0[4 * q + e] * inv); h1[e] = (half_t)(o1[4 * q + e] * inv); }
                *(half4 *)(o + d) = h0;
                *(half4 *)(o + 32 + d) = h1;
            }
        }
    }
}

__device__ __forceinline__ void att_glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l,
                                     16, 0, 0);
}

// Short utterances (T <= 128 encoder frames: every 10 s clip of the headline batch is 126).  The key-tiled kernel below
// walks the keys tile by tile with an online softmax, which for four key tiles is four times (stage -> barrier ->
// fragments -> MFMA -> skew through LDS -> softmax -> MFMA) in a row: 16 us of dependent chain for 0.5 us of MFMA.
// Here a block is one (utterance, head): its four waves (one query tile each) stage ALL of K, V^T and the 256
// relative-position rows the four query tiles can touch with coalesced direct-to-LDS loads (16 wave-loads per wave, one
// latency, one barrier; a first version that fetched the fragments straight from global memory -- 56 scattered loads per
// wave, every wave of the block fetching the same K and V -- ran 25 us, bound by the vector-memory pipeline), then each
// wave issues the K / position products of the WHOLE key range back to back (<= 16 + 20 MFMAs: the five position tiles
// of its 160-row window are each computed once, not once per key tile that touches them), skews tile after tile through
// its LDS slab (which takes the place of K and the position rows after a second barrier), takes ONE softmax over the
// complete row (no running maximum, no rescaling of the output accumulator) and ends with the <= 16 P.V products.
// 64 KB of LDS and < 256 registers: two blocks per CU, the headline batch (512 blocks) in one round.  Not the same
// summation as the key-tiled kernels (single-pass softmax), so not bit-identical to them; WHICH kernel an utterance gets
// depends on its own length only, never on the batch around it, so an utterance alone and in any batch has the same bits.
#define ATT_SHORT_T 128
#ifdef QV_ATT_STAMPS   // tools/att_bench.hip: shader-clock stamps of one block's waves at the phase boundaries
__device__ long long g_att_stamp[4][8];
#define ATT_STAMP(i) do { if (blockIdx.x == 3 && blockIdx.y == 17 && lane == 0) g_att_stamp[wave][i] = clock64(); } while (0)
#else
#define ATT_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_attention_short(
    const half_t *__restrict__ qk, const half_t *__restrict__ vt, const half_t *__restrict__ pos, int pos_ld,
    const float *__restrict__ bias_u, const float *__restrict__ bias_v, const int32_t *__restrict__ len,
    const int32_t *__restrict__ row_off, half_t *__restrict__ out, int t_max, int t_pad) {
    // [0, 16 K): K as 4 tiles [32 keys][64], 16-B chunks swizzled by (key >> 1) & 7; [16 K, 48 K): 256 position rows [64],
    // swizzled the same way; [48 K, 64 K): V^T as 4 tiles [64 d][32 keys], chunks swizzled by (d >> 2) & 3.
    // The four skew slabs (34,304 B) reuse [0, 48 K) once every wave has its K / position products.
    __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
    half_t *sK = (half_t *)smem, *sP = (half_t *)(smem + 16 * 1024), *sV = (half_t *)(smem + 48 * 1024);
    // (round 5: handing XCD k the k-th eighth of the (utterance, head) pairs -- xcd_order, as in the LayerNorms -- was
    // measured: 16.46 -> 16.72 us; the kernel is not waiting for its K / V rows.  Kept head-major.)
    const int b = blockIdx.y, h = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int T = len[b];
    if (T > ATT_SHORT_T) return;                // whole block: a long utterance belongs to k_attention_ws
    ATT_STAMP(0);
    const size_t row0 = (size_t)row_off[b];
    const int l31 = lane & 31, hi = lane >> 5;
    const int i0 = wave * 32;
    const bool active = i0 < T;
    const half_t *qb = qk + row0 * (2 * QV_D) + h * QV_DK;
    const half_t *kb = qb + QV_D;
    const half_t *vb = vt + ((size_t)b * QV_D + h * QV_DK) * t_pad;
    const half_t *pb = pos + h * QV_DK;
    const int n_kt = (T + 31) >> 5;             // 1..4 key tiles = query tiles
    const int Rb = t_max - 128;                 // position row of window row 0
    // the query rows first: their latency is the longest chain in front of the first product (convert, add the biases)
    half8 q8[4];
    {
        int qi = i0 + l31;
        qi = qi < T ? qi : T - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) q8[ks] = *(const half8 *)(qb + (size_t)qi * (2 * QV_D) + ks * 16 + hi * 8);
    }
    // ---- stage: wave w loads keys 8 w .. 8 w + 7 of every key tile, d rows 16 w .. of every V^T tile, its share of the
    // position rows [96 - 32 (n_kt - 1), 96 + 32 (n_kt + 1)) that the active query tiles touch
    for (int kt = 0; kt < n_kt; ++kt) {
        {
            int r = wave * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
            int kj = kt * 32 + r;
            kj = kj < T ? kj : T - 1;
            att_glds16(kb + (size_t)kj * (2 * QV_D) + c * 8, sK + kt * 2048 + wave * 512);
        }
        {
            int r = wave * 16 + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
            att_glds16(vb + (size_t)r * t_pad + kt * 32 + c * 8, sV + kt * 2048 + wave * 512);
        }
    }
    for (int q = 4 * (4 - n_kt) + wave; q < 4 * (4 + n_kt); q += 4) {      // chunks of 8 window rows
        int wr = q * 8 + (lane >> 3);
        int c = (lane & 7) ^ ((wr >> 1) & 7);
        int rr = Rb + wr;
        rr = rr < 0 ? 0 : (rr > 2 * t_max - 2 ? 2 * t_max - 2 : rr);
        att_glds16(pb + (size_t)rr * pos_ld + c * 8, sP + q * 512);
    }
    half8 qu[4], qv[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        int d = ks * 16 + hi * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float qf = (float)q8[ks][e];
            qu[ks][e] = (half_t)(qf + bias_u[h * QV_DK + d + e]);
            qv[ks][e] = (half_t)(qf + bias_v[h * QV_DK + d + e]);
        }
    }
    ATT_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ATT_STAMP(2);
    // S^T[kt][jj][ii] for every key tile, raw^T[pt][c][ii] for position tiles pt = 0 .. n_kt (window rows 96 - i0 + 32 pt + c)
    f32x16 st[4], raw[5];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
        if (active && kt < n_kt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = ks * 2 + hi;
                half8 kf = *(const half8 *)(sK + kt * 2048 + l31 * 64 + ((c ^ ((l31 >> 1) & 7)) << 3));
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qu[ks], st[kt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int pt = 0; pt < 5; ++pt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) raw[pt][r] = 0.f;
        if (active && pt <= n_kt) {
            const int wr = 96 - i0 + 32 * pt + l31;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = ks * 2 + hi;
                half8 pf = *(const half8 *)(sP + wr * 64 + ((c ^ ((wr >> 1) & 7)) << 3));
                raw[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf, qv[ks], raw[pt], 0, 0, 0);
            }
        }
    }
    __syncthreads();                            // K and the position rows are dead: the slabs take their place
    if (!active) return;
    ATT_STAMP(3);
    float *sl = (float *)smem + wave * (32 * ATT_LDS_LD);
    // skew tile by tile: BD^T of key tile kt is slab[ii][31 - ii + jj] over the 64 columns (raw^T tile kt | tile kt + 1).
    // Every position tile is written ONCE: tile pt lives in slab columns 32 (pt & 1) .., so key tile kt finds tile kt in
    // one half and only tile kt + 1 has to replace tile kt - 1 in the other; the read column wraps modulo 64.
    // st[kt] becomes the masked, scaled score
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sl[l31 * ATT_LDS_LD + (r & 3) + 8 * (r >> 2) + 4 * hi] = raw[0][r];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        if (kt < n_kt) {
            __builtin_amdgcn_wave_barrier();    // the reads of key tile kt - 1 are issued before tile kt + 1 lands on tile kt - 1
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sl[l31 * ATT_LDS_LD + 32 * ((kt + 1) & 1) + (r & 3) + 8 * (r >> 2) + 4 * hi] = raw[kt + 1][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int jj = (r & 3) + 8 * (r >> 2) + 4 * hi;
                float bd = sl[l31 * ATT_LDS_LD + ((32 * (kt & 1) + 31 - l31 + jj) & 63)];
                float sc = (st[kt][r] + bd) * 0.125f;
                sc = (kt * 32 + jj < T) ? sc : -1e30f;
                st[kt][r] = sc;
                mx = fmaxf(mx, sc);
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    ATT_STAMP(4);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        if (kt < n_kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int jj = (r & 3) + 8 * (r >> 2) + 4 * hi;
                float e = (kt * 32 + jj < T) ? __expf(st[kt][r] - mx) : 0.f;
                st[kt][r] = e;
                sum += e;
            }
        }
    }
    sum += __shfl_xor(sum, 32);
    ATT_STAMP(5);
    // P.V: B operand = exp(S^T) registers 8ks .. 8ks + 7 of tile kt (keys 32 kt + 16 ks + 4 hi + {0..3, 8..11});
    // A operand = V^T rows d with the same key order: 16-B chunks 2 ks and 2 ks + 1 of the tile, 8-byte half `hi`
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        if (kt < n_kt) {
            const half_t *sVb = sV + kt * 2048;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 pbf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pbf[e] = (half_t)st[kt][8 * ks + e];
                const int da = l31, dc = 32 + l31;
                half4 a0 = *(const half4 *)(sVb + da * 32 + (((2 * ks) ^ ((da >> 2) & 3)) << 3) + 4 * hi);
                half4 a1 = *(const half4 *)(sVb + da * 32 + (((2 * ks + 1) ^ ((da >> 2) & 3)) << 3) + 4 * hi);
                half4 c0 = *(const half4 *)(sVb + dc * 32 + (((2 * ks) ^ ((dc >> 2) & 3)) << 3) + 4 * hi);
                half4 c1 = *(const half4 *)(sVb + dc * 32 + (((2 * ks + 1) ^ ((dc >> 2) & 3)) << 3) + 4 * hi);
                half8 va = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                half8 vc = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pbf, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vc, pbf, o1, 0, 0, 0);
            }
        }
    }
    if (i0 + l31 < T) {
        float inv = 1.f / sum;
        half_t *o = out + (row0 + i0 + l31) * QV_D + h * QV_DK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int d = 8 * q + 4 * hi;
            half4 h0, h1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { h0[e] = (half_t)(o0[4 * q + e] * inv); h1[e] = (half_t)(o1[4 * q + e] * inv); }
            *(half4 *)(o + d) = h0;
            *(half4 *)(o + 32 + d) = h1;
        }
    }
    ATT_STAMP(6);
}

// Wave-specialised version of the same computation (same arithmetic, same order).  k_attention's
// duration is one wave's serial chain -- per key tile two exposed global-load latencies (K and
// position fragments, then V) in front of 16 MFMAs -- and it keeps a wave busy with one query tile
// after the other for long utterances.  Here a block is one (head, utterance, group of 4 query
// tiles): waves 0..3 are consumers (one query tile each; fragments come from LDS), waves 4..7 are
// loaders that stage, TWO key tiles ahead (3 buffers), the K tile (32 keys x 64), the V^T tile
// (64 x 32 keys) and the 32 NEW relative-position rows of the tile: the position rows the four query
// tiles need form a sliding window of 160 rows that advances by 32 per key tile, kept in a 256-row
// ring (window of tile kt + the 64 rows prefetched for kt + 1 and kt + 2 never overlap in it).
// HPB heads share a block (4 consumer waves each; the 4 loader waves stage for all of them): with HPB = 2 every SIMD runs
// two consumer waves whose serial chains (MFMA -> skew through LDS -> softmax -> MFMA) overlap, and B = 64 x 10 s is one
// round of 256 blocks instead of two rounds of 512.  LDS then only holds NST = 2 K/V stages (tile kt + 1 is requested
// when tile kt's barrier frees the other buffer) and a 192-row position ring (160-row window + the 32 rows of the next
// tile).  The arithmetic of a (head, query tile) is the same wave program in both shapes: outputs are bit-identical.

#ifdef QV_ATT_STAMPS   // tools/att_bench.hip: per key tile, clock stamps of every wave of ONE block of the key-tiled kernel
__device__ long long g_ws_stamp[12][16][6];
#define WS_STAMP(kt, i) do { if (blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 17 && lane == 0 && (kt) < 16) g_ws_stamp[wave][kt][i] = clock64(); } while (0)
#else
#define WS_STAMP(kt, i) do { } while (0)
#endif
template <int HPB, int NST, int RING>
__global__ __launch_bounds__(256 * HPB + 256) void k_attention_ws(const half_t *__restrict__ qk, const half_t *__restrict__ vt,
                                                                  const half_t *__restrict__ pos, int pos_ld,
                                                                  const float *__restrict__ bias_u, const float *__restrict__ bias_v,
                                                                  const int32_t *__restrict__ len, const int32_t *__restrict__ row_off,
                                                                  half_t *__restrict__ out, int t_max, int t_pad, int t_short) {
    static_assert(RING % 8 == 0 && RING >= 160 + 32 * (NST - 1), "position ring: window + prefetched rows");
    __shared__ __attribute__((aligned(16))) half_t sK[HPB][NST][32 * 64];    // [key][d], 16-B chunks swizzled by (key >> 1) & 7
    __shared__ __attribute__((aligned(16))) half_t sV[HPB][NST][64 * 32];    // [d][key], 16-B chunks swizzled by (d >> 2) & 3
    __shared__ __attribute__((aligned(16))) half_t sP[HPB][RING * 64];       // ring of position rows, swizzled like sK
    __shared__ float slab[4 * HPB][32 * ATT_LDS_LD];
    const int b = blockIdx.z, qg = blockIdx.y, hg = blockIdx.x;
    const int T = len[b];
    if (qg * 128 >= T || T <= t_short) return;       // whole block: no query of this group exists / k_attention_short's utterance
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool loader = wave >= 4 * HPB;
    const int w4 = wave & 3, l31 = lane & 31, hi = lane >> 5;
    const size_t row0 = (size_t)row_off[b];
    const int n_kt = (T + 31) >> 5;
    const int Rb = t_max - 128 * qg - 128;           // first position row of key tile 0's window

    if (loader) {
        // per head: one K, one V and one position load per loader wave and key tile (1 KB each)
        auto stage = [&](int kt) {
            const int j0 = kt * 32, buf = kt % NST;
#pragma unroll
            for (int hh = 0; hh < HPB; ++hh) {
                const int h = hg * HPB + hh;
                const half_t *kb = qk + row0 * (2 * QV_D) + QV_D + h * QV_DK;
                const half_t *vb = vt + ((size_t)b * QV_D + h * QV_DK) * t_pad;
                {   // K: 8 keys x 128 B per wave
                    int r = w4 * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
                    int kj = j0 + r;
                    kj = kj < T ? kj : T - 1;
                    att_glds16(kb + (size_t)kj * (2 * QV_D) + c * 8, sK[hh][buf] + w4 * 512);
                }
                {   // V^T: 16 d rows x 64 B per wave (keys j0 .. j0 + 31 < t_pad)
                    int r = w4 * 16 + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
                    att_glds16(vb + (size_t)r * t_pad + j0 + c * 8, sV[hh][buf] + w4 * 512);
                }
            }
        };
        auto pos_rows = [&](int first, int n8) {   // n8 chunks of 8 rows starting at window row `first`, this wave's share
            for (int q = w4; q < n8; q += 4) {
                int wr = first + q * 8 + (lane >> 3);                 // row relative to Rb
                int ring = wr % RING;
                int c = (lane & 7) ^ ((ring >> 1) & 7);
                int rr = Rb + wr;
                rr = rr < 0 ? 0 : (rr > 2 * t_max - 2 ? 2 * t_max - 2 : rr);
#pragma unroll
                for (int hh = 0; hh < HPB; ++hh)
                    att_glds16(pos + (hg * HPB + hh) * QV_DK + (size_t)rr * pos_ld + c * 8,
                               sP[hh] + (size_t)((first + q * 8) % RING) * 64);
            }
        };
        // per wave: unit(0) = HPB x (5 position chunks + K + V), unit(kt >= 1) = HPB x (K + V + 1 position chunk)
        pos_rows(0, 20);        // rows 0 .. 159: the window of tile 0
        stage(0);
        if (NST == 3 && n_kt > 1) {
            stage(1);
            pos_rows(32 + 128, 4);
        }
        for (int kt = 0; kt < n_kt; ++kt) {
            WS_STAMP(kt, 0);
            // unit(kt) has landed once only the units requested after it may still be in flight
            if (NST == 3 && kt + 1 < n_kt) {
                if (HPB == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WS_STAMP(kt, 1);
            __builtin_amdgcn_s_barrier();           // tile kt visible; the buffers of tile kt - 1 are free
            WS_STAMP(kt, 2);
            if (kt + NST - 1 < n_kt) {
                stage(kt + NST - 1);
                pos_rows(32 * (kt + NST - 1) + 128, 4);   // the 32 new rows of that tile
            }
            WS_STAMP(kt, 3);
        }
        return;
    }

    // ---------------------------------------------------------------- consumers ----------
    const int hh = wave >> 2, h = hg * HPB + hh;
    const int i0 = qg * 128 + w4 * 32;
    const bool active = i0 < T;
    const half_t *qb = qk + row0 * (2 * QV_D) + h * QV_DK;
    float *sl = slab[wave];
    const half_t *sPh = sP[hh];
    half8 qu[4], qv[4];
    {
        int qi = i0 + l31;
        qi = qi < T ? qi : T - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            int d = ks * 16 + hi * 8;
            half8 q8 = *(const half8 *)(qb + (size_t)qi * (2 * QV_D) + d);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float qf = (float)q8[e];
                qu[ks][e] = (half_t)(qf + bias_u[h * QV_DK + d + e]);
                qv[ks][e] = (half_t)(qf + bias_v[h * QV_DK + d + e]);
            }
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    // Round 6.  The loop below is bound by the instructions the two consumer waves of a SIMD issue per key tile (stamps in
    // tools/att_bench: 4.4 k cycles per tile, the loaders land a tile ahead and wait), so the same arithmetic is issued
    // with fewer of them -- the output bits do not change (tests compare variants 0 .. 2):
    //  * scores stay in RAW units (8 x the scaled score) up to the exponential: exp((p - m) / 8) = exp2((p - m) * (log2 e / 8)),
    //    and scaling by a power of two commutes with every rounding on the way -- 16 multiplies per tile gone;
    //  * the key mask (j0 + jj < T) only exists in an utterance's LAST key tile: the other tiles run an unmasked copy of
    //    the body (16 compares + 32 selects gone);
    //  * the skew slab keeps alternating halves (tile kt sits where tile kt - 1 wrote it as ITS second tile), but the column
    //    of a value is no longer computed with a wrap: an even key tile reads lane constant + immediate (31 - ii + jj <= 62
    //    never wraps), an odd one reads the same column XOR 32, i.e. one of two lane-constant bases chosen by a compare of two
    //    constants per value -- ~5 integer instructions per value read become 0 / 2;
    //  * fragment addresses: lane constants computed once, XORed with the K-step.
    constexpr float NEG = -1e30f * 8.0f, L2E8 = 0x1.715476p+0f * 0.125f;   // (__expf(x) is v_exp_f32(x * 0x1.715476p+0))
    float m_run = NEG, l_run = 0.f;
    const int swz = (l31 >> 1) & 7;
    const int kf0 = l31 * 64 + ((hi ^ swz) << 3);                  // K / position fragment of K-step ks: kf0 ^ (ks << 4) (halves)
    float *const slw = sl + l31 * ATT_LDS_LD + 4 * hi;            // slab stores: + 32 * tile + (r & 3) + 8 * (r >> 2)
    const float *const slr = sl + l31 * ATT_LDS_LD + (31 - l31 + 4 * hi);   // slab loads: + (r & 3) + 8 * (r >> 2), never past column 62
    // odd tiles: slab column = window column ^ 32, i.e. + 32 where the window column 31 - ii + jj is below 32 and - 32 elsewhere;
    // bit r of `hiw` says "elsewhere" for register r, so the address is one shift-add away from a lane constant
    unsigned hiw = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) hiw |= (unsigned)((r & 3) + 8 * (r >> 2) >= 1 + l31 - 4 * hi) << r;
    const int vda = l31 * 32 + 4 * hi, vdc = (32 + l31) * 32 + 4 * hi, vsw = (l31 >> 2) & 3;   // V^T rows d = l31 and 32 + l31 share (d >> 2) & 3
    int ring1 = (128 - 32 * w4) % RING;                          // position tile kt + 1 of this wave starts at ring row 32 kt + 128 - 32 w4
    auto tile = [&](int kt, auto masked_c, auto odd_c) {
        constexpr bool MASKED = decltype(masked_c)::value, ODD = decltype(odd_c)::value;   // ODD: kt & 1 (never kt == 0)
        const int j0 = kt * 32, buf = kt % NST;
        // (the per-K-step fragment offsets are re-derived from three lane constants in every tile -- one XOR / shift each: hoisted
        // out of the loop they are 14 more live registers, and the two-heads-per-block shape has 168 per wave)
        int kfl = kf0, val = vda, vcl = vdc;
        asm volatile("" : "+v"(kfl), "+v"(val), "+v"(vcl));
        const half_t *sKb = sK[hh][buf], *sVb = sV[hh][buf];
        // position tile pt = rows 32 pt .. of the wave's window: key tile kt needs tiles kt and kt + 1, and tile kt is what
        // key tile kt - 1 computed as ITS second tile -- same operands, same products, so it is carried over instead of
        // being computed twice
        f32x16 st, r1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; r1[r] = 0.f; }
        if (!ODD && kt == 0) {
            const int ring0 = (96 - 32 * w4) % RING;
            f32x16 r0;
#pragma unroll
            for (int r = 0; r < 16; ++r) r0[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                half8 p0 = *(const half8 *)(sPh + ring0 * 64 + (kfl ^ (ks << 4)));
                r0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, qv[ks], r0, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) slw[(r & 3) + 8 * (r >> 2)] = r0[r];
        }
        const half_t *sP1 = sPh + ring1 * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            half8 kf = *(const half8 *)(sKb + (kfl ^ (ks << 4)));
            half8 p1 = *(const half8 *)(sP1 + (kfl ^ (ks << 4)));
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qu[ks], st, 0, 0, 0);
            r1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, qv[ks], r1, 0, 0, 0);
        }
        ring1 += 32;
        if (ring1 >= RING) ring1 -= RING;
        // skew: BD^T[jj][ii] = window[ii][31 - ii + jj], window = [tile kt | tile kt + 1]; tile pt lives in slab columns
        // 32 (pt & 1) .., so tile kt is already there and tile kt + 1 replaces tile kt - 1
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) slw[(ODD ? 0 : 32) + (r & 3) + 8 * (r >> 2)] = r1[r];
        __builtin_amdgcn_wave_barrier();
        WS_STAMP(kt, 2);
        float p[16];
        float mx = NEG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jj = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int e = (r & 3) + 8 * (r >> 2);
            // even tile: window column = slab column; odd tile: slab column = window column ^ 32
            const float bd = !ODD ? slr[e] : *(const float *)((const char *)(slr + 32 + e) - (((hiw >> r) & 1u) << 8));
            const float sc = st[r] + bd;
            p[r] = (!MASKED || j0 + jj < T) ? sc : NEG;
            mx = fmaxf(mx, p[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float corr = __builtin_amdgcn_exp2f((m_run - m_new) * L2E8);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jj = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float e = (!MASKED || j0 + jj < T) ? __builtin_amdgcn_exp2f((p[r] - m_new) * L2E8) : 0.f;
            p[r] = e;
            sum += e;
        }
        sum += __shfl_xor(sum, 32);
        l_run = l_run * corr + sum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }
        WS_STAMP(kt, 3);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 pbf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pbf[e] = (half_t)p[8 * ks + e];
            // V^T pieces: keys 16 ks + 4 hi + {0..3} and + 8 -> 16-B chunks 2 ks and 2 ks + 1, 8-byte half `hi`
            half4 a0 = *(const half4 *)(sVb + val + (((2 * ks) ^ vsw) << 3));
            half4 a1 = *(const half4 *)(sVb + val + (((2 * ks + 1) ^ vsw) << 3));
            half4 c0 = *(const half4 *)(sVb + vcl + (((2 * ks) ^ vsw) << 3));
            half4 c1 = *(const half4 *)(sVb + vcl + (((2 * ks + 1) ^ vsw) << 3));
            half8 v0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            half8 v1 = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pbf, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pbf, o1, 0, 0, 0);
        }
        WS_STAMP(kt, 4);
    };
    for (int kt = 0; kt < n_kt; ++kt) {
        WS_STAMP(kt, 0);
        __builtin_amdgcn_s_barrier();
        WS_STAMP(kt, 1);
        if (!active) continue;
        if (kt + 1 < n_kt) { if (kt & 1) tile(kt, std::false_type{}, std::true_type{}); else tile(kt, std::false_type{}, std::false_type{}); }
        else { if (kt & 1) tile(kt, std::true_type{}, std::true_type{}); else tile(kt, std::true_type{}, std::false_type{}); }
    }
    if (active && i0 + l31 < T) {
        float inv = 1.f / l_run;
        half_t *o = out + (row0 + i0 + l31) * QV_D + h * QV_DK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int d = 8 * q + 4 * hi;
            half4 h0, h1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { h0[e] = (half_t)(o0[4 * q + e] * inv); h1[e] = (half_t)(o1[4 * q + e] * inv); }
            *(half4 *)(o + d) = h0;
            *(half4 *)(o + 32 + d) = h1;
        }
    }
}

// Round 6, variant 4: the key-tiled computation WITHOUT loader waves and with TWO independent blocks per CU.
// k_attention_ws<2, 2, 192> is bound by the instruction issue of the two consumer waves of a SIMD, and those two run in lock
// step: both heads of a block leave the same barrier, so both are in their MFMA phase, then both in their softmax phase
// (stamps: tools/att_bench).  Here a block is ONE (head, 128-query group, utterance) with four waves that stage their own
// tiles -- each wave issues its quarter of the K, V^T and position loads of key tile kt + 1 (three 1 KB direct-to-LDS loads)
// right after the barrier that released tile kt -- in 74 KB of LDS and, without loader waves, 256 threads: two blocks per
// CU, 256 registers per wave.  The two consumer waves of a SIMD then belong to DIFFERENT blocks, whose barriers are
// independent: one wave's matrix products run under the other's softmax.  Same wave program per (head, query tile) as the
// other key-tiled shapes => the same bits (tests/test_gpu_forward.py compares variants 0, 1, 2 and 4).
template <int RING>
__global__ __launch_bounds__(256, 2) void k_attention_x(const half_t *__restrict__ qk, const half_t *__restrict__ vt,
                                                         const half_t *__restrict__ pos, int pos_ld,
                                                         const float *__restrict__ bias_u, const float *__restrict__ bias_v,
                                                         const int32_t *__restrict__ len, const int32_t *__restrict__ row_off,
                                                         half_t *__restrict__ out, int t_max, int t_pad, int t_short) {
    static_assert(RING % 32 == 0 && RING >= 160 + 32, "position ring: window + the rows of the next tile");
    __shared__ __attribute__((aligned(16))) half_t sK[2][32 * 64];      // [key][d], 16-B chunks swizzled by (key >> 1) & 7
    __shared__ __attribute__((aligned(16))) half_t sV[2][64 * 32];      // [d][key], 16-B chunks swizzled by (d >> 2) & 3
    __shared__ __attribute__((aligned(16))) half_t sP[RING * 64];       // ring of position rows, swizzled like sK
    __shared__ float slab[4][32 * ATT_LDS_LD];
    const int b = blockIdx.z, qg = blockIdx.y, h = blockIdx.x;
    const int T = len[b];
    if (qg * 128 >= T || T <= t_short) return;
    const int w4 = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const size_t row0 = (size_t)row_off[b];
    const int n_kt = (T + 31) >> 5;
    const int Rb = t_max - 128 * qg - 128;           // first position row of key tile 0's window
    const half_t *kb = qk + row0 * (2 * QV_D) + QV_D + h * QV_DK;
    const half_t *vb = vt + ((size_t)b * QV_D + h * QV_DK) * t_pad;
    const half_t *pb = pos + h * QV_DK;
    auto stage = [&](int kt) {                        // this wave's quarter of key tile kt: 8 keys of K, 16 d rows of V^T
        const int j0 = kt * 32, buf = kt & 1;
        {
            int r = w4 * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
            int kj = j0 + r;
            kj = kj < T ? kj : T - 1;
            att_glds16(kb + (size_t)kj * (2 * QV_D) + c * 8, sK[buf] + w4 * 512);
        }
        {
            int r = w4 * 16 + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
            att_glds16(vb + (size_t)r * t_pad + j0 + c * 8, sV[buf] + w4 * 512);
        }
    };
    auto pos_rows = [&](int first, int n8) {          // n8 chunks of 8 rows starting at window row `first`, this wave's share
        for (int q = w4; q < n8; q += 4) {
            int wr = first + q * 8 + (lane >> 3);
            int ring = wr % RING;
            int c = (lane & 7) ^ ((ring >> 1) & 7);
            int rr = Rb + wr;
            rr = rr < 0 ? 0 : (rr > 2 * t_max - 2 ? 2 * t_max - 2 : rr);
            att_glds16(pb + (size_t)rr * pos_ld + c * 8, sP + (size_t)((first + q * 8) % RING) * 64);
        }
    };
    pos_rows(0, 20);                                  // rows 0 .. 159: the window of tile 0
    stage(0);

    const int i0 = qg * 128 + w4 * 32;
    const bool active = i0 < T;
    const half_t *qb = qk + row0 * (2 * QV_D) + h * QV_DK;
    float *sl = slab[w4];
    half8 qu[4], qv[4];
    {
        int qi = i0 + l31;
        qi = qi < T ? qi : T - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            int d = ks * 16 + hi * 8;
            half8 q8 = *(const half8 *)(qb + (size_t)qi * (2 * QV_D) + d);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float qf = (float)q8[e];
                qu[ks][e] = (half_t)(qf + bias_u[h * QV_DK + d + e]);
                qv[ks][e] = (half_t)(qf + bias_v[h * QV_DK + d + e]);
            }
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    // (the wave program below is k_attention_ws's, statement for statement: raw-unit softmax, mask in the last tile only,
    // alternating slab halves with wrap-free addressing)
    constexpr float NEG = -1e30f * 8.0f, L2E8 = 0x1.715476p+0f * 0.125f;
    float m_run = NEG, l_run = 0.f;
    const int swz = (l31 >> 1) & 7;
    const int kf0 = l31 * 64 + ((hi ^ swz) << 3);
    float *const slw = sl + l31 * ATT_LDS_LD + 4 * hi;
    const float *const slr = sl + l31 * ATT_LDS_LD + (31 - l31 + 4 * hi);
    unsigned hiw = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) hiw |= (unsigned)((r & 3) + 8 * (r >> 2) >= 1 + l31 - 4 * hi) << r;
    const int vda = l31 * 32 + 4 * hi, vdc = (32 + l31) * 32 + 4 * hi, vsw = (l31 >> 2) & 3;
    int ring1 = (128 - 32 * w4) % RING;
    auto tile = [&](int kt, auto masked_c, auto odd_c) {
        constexpr bool MASKED = decltype(masked_c)::value, ODD = decltype(odd_c)::value;
        const int j0 = kt * 32, buf = kt & 1;
        const half_t *sKb = sK[buf], *sVb = sV[buf];
        f32x16 st, r1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; r1[r] = 0.f; }
        if (!ODD && kt == 0) {
            const int ring0 = (96 - 32 * w4) % RING;
            f32x16 r0;
#pragma unroll
            for (int r = 0; r < 16; ++r) r0[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                half8 p0 = *(const half8 *)(sP + ring0 * 64 + (kf0 ^ (ks << 4)));
                r0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, qv[ks], r0, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) slw[(r & 3) + 8 * (r >> 2)] = r0[r];
        }
        const half_t *sP1 = sP + ring1 * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            half8 kf = *(const half8 *)(sKb + (kf0 ^ (ks << 4)));
            half8 p1 = *(const half8 *)(sP1 + (kf0 ^ (ks << 4)));
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qu[ks], st, 0, 0, 0);
            r1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, qv[ks], r1, 0, 0, 0);
        }
        ring1 += 32;
        if (ring1 >= RING) ring1 -= RING;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) slw[(ODD ? 0 : 32) + (r & 3) + 8 * (r >> 2)] = r1[r];
        __builtin_amdgcn_wave_barrier();
        float p[16];
        float mx = NEG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jj = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int e = (r & 3) + 8 * (r >> 2);
            const float bd = !ODD ? slr[e] : *(const float *)((const char *)(slr + 32 + e) - (((hiw >> r) & 1u) << 8));
            const float sc = st[r] + bd;
            p[r] = (!MASKED || j0 + jj < T) ? sc : NEG;
            mx = fmaxf(mx, p[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float corr = __builtin_amdgcn_exp2f((m_run - m_new) * L2E8);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jj = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float e = (!MASKED || j0 + jj < T) ? __builtin_amdgcn_exp2f((p[r] - m_new) * L2E8) : 0.f;
            p[r] = e;
            sum += e;
        }
        sum += __shfl_xor(sum, 32);
        l_run = l_run * corr + sum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 pbf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pbf[e] = (half_t)p[8 * ks + e];
            half4 a0 = *(const half4 *)(sVb + vda + (((2 * ks) ^ vsw) << 3));
            half4 a1 = *(const half4 *)(sVb + vda + (((2 * ks + 1) ^ vsw) << 3));
            half4 c0 = *(const half4 *)(sVb + vdc + (((2 * ks) ^ vsw) << 3));
            half4 c1 = *(const half4 *)(sVb + vdc + (((2 * ks + 1) ^ vsw) << 3));
            half8 v0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            half8 v1 = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pbf, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pbf, o1, 0, 0, 0);
        }
    };
    for (int kt = 0; kt < n_kt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's quarter of tile kt (and of its position rows) has landed ...
        __builtin_amdgcn_s_barrier();                        // ... and everybody's; everybody has left tile kt - 1's buffers
        if (kt + 1 < n_kt) {
            stage(kt + 1);
            pos_rows(32 * (kt + 1) + 128, 4);               // the 32 new rows of that tile
        }
        if (!active) continue;
        if (kt + 1 < n_kt) { if (kt & 1) tile(kt, std::false_type{}, std::true_type{}); else tile(kt, std::false_type{}, std::false_type{}); }
        else { if (kt & 1) tile(kt, std::true_type{}, std::true_type{}); else tile(kt, std::true_type{}, std::false_type{}); }
    }
    if (active && i0 + l31 < T) {
        float inv = 1.f / l_run;
        half_t *o = out + (row0 + i0 + l31) * QV_D + h * QV_DK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int d = 8 * q + 4 * hi;
            half4 h0, h1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { h0[e] = (half_t)(o0[4 * q + e] * inv); h1[e] = (half_t)(o1[4 * q + e] * inv); }
            *(half4 *)(o + d) = h0;
            *(half4 *)(o + 32 + d) = h1;
        }
    }
}

// ------------------------------------------------------------------ conv module --------
// depthwise Conv1d(512, k=9, pad 4) + folded BatchNorm + Swish on f16 [M][512]; input frames
// t >= len[b] read as zero (the reference zeroes padded frames after GLU).  A lane owns 8
// channels (weights [9][512] tap-major in registers) and slides over DW_TT consecutive frames,
// so each input row is loaded once per DW_TT outputs instead of 9 times.
#define DW_TT 4
__global__ __launch_bounds__(256) void k_dwconv1d(const half_t *__restrict__ x, const float *__restrict__ wt /*[9][512]*/,
                                                  const float *__restrict__ bias, const int32_t *__restrict__ len,
                                                  const int32_t *__restrict__ row_off, half_t *__restrict__ y) {
    // blocks in (utterance, frame chunk) order, XCD k takes the k-th eighth of them (xcd_order)
    const int wg = xcd_order(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int b = wg / gridDim.x, t0 = ((wg - b * gridDim.x) * 4 + (threadIdx.x >> 6)) * DW_TT, lane = threadIdx.x & 63;
    const int T = len[b], c0 = lane * 8;
    if (t0 >= T) return;
    const size_t row0 = (size_t)row_off[b];   // packed rows of utterance b
    float w[9][8], bs[8];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        f32x4 w0 = *(const f32x4 *)(wt + k * QV_D + c0), w1 = *(const f32x4 *)(wt + k * QV_D + c0 + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { w[k][c] = w0[c]; w[k][4 + c] = w1[c]; }
    }
    {
        f32x4 b0 = *(const f32x4 *)(bias + c0), b1 = *(const f32x4 *)(bias + c0 + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { bs[c] = b0[c]; bs[4 + c] = b1[c]; }
    }
    float acc[DW_TT][8];
#pragma unroll
    for (int j = 0; j < DW_TT; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[j][c] = bs[c];
#pragma unroll
    for (int i = 0; i < DW_TT + 8; ++i) {
        int tt = t0 - 4 + i;
        if (tt < 0 || tt >= T) continue;
        half8 v = *(const half8 *)(x + (row0 + tt) * QV_D + c0);
        float vf[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) vf[c] = (float)v[c];
#pragma unroll
        for (int j = 0; j < DW_TT; ++j) {
            int k = i - j;  // tap index: tt = (t0 + j) + k - 4
            if (k < 0 || k > 8) continue;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[j][c] = __builtin_fmaf(w[k][c], vf[c], acc[j][c]);
        }
    }
#pragma unroll
    for (int j = 0; j < DW_TT; ++j) {
        if (t0 + j >= T) break;
        half8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = (half_t)(acc[j][c] * sigmoidf_(acc[j][c]));
        *(half8 *)(y + (row0 + t0 + j) * QV_D + c0) = o;
    }
}

// ------------------------------------------------------------------ log-softmax --------
// logits f32 [M][ld] (first 1025 valid) -> log-probs f32 [B][t_max][1025]; one wave per row.
__global__ __launch_bounds__(256) void k_logsoftmax(const float *__restrict__ logits, int ld, float *__restrict__ out, int M,
                                                    const int32_t *__restrict__ row_map, int t_out) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    // packed row -> (utterance, frame); the caller's log-prob tensor is dense [B][t_out][1025]
    const int bt = row_map[row];
    const size_t orow = (size_t)(bt >> 16) * t_out + (bt & 0xFFFF);
    const float *p = logits + (size_t)row * ld;
    float v[17];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        int c = lane + 64 * i;
        v[i] = c < 1025 ? p[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 17; ++i) s += (lane + 64 * i < 1025) ? expf(v[i] - mx) : 0.f;
    float lse = mx + logf(wave_sum(s));
    float *o = out + orow * 1025;
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        int c = lane + 64 * i;
        if (c < 1025) o[c] = v[i] - lse;
    }
}

// rows t >= T[b] of the caller's dense [B][t_out][1025] tensor: zeros (qv_forward's contract -- onnxruntime hands the
// reference exactly [1, T, 1025], mixed/run.py:59-63; a padded batch tensor must not expose uninitialised memory)
__global__ __launch_bounds__(256) void k_zero_pad_rows(float *__restrict__ out, const int32_t *__restrict__ len, int t_out) {
    const int b = blockIdx.y, t = len[b] + blockIdx.x;
    if (t >= t_out) return;
    float *o = out + ((size_t)b * t_out + t) * 1025;
    for (int c = threadIdx.x; c < 1025; c += 256) o[c] = 0.f;
}

}  // namespace

// ====================================================================== launchers ======

// stats: [batch][80][2] results, followed by room for the [batch][MS_CHUNKS][80][2] partial sums (qv_melstats_doubles)
size_t qv_melstats_doubles(size_t max_batch) { return max_batch * 2 * QV_NMEL * (1 + MS_CHUNKS); }
static std::atomic<int> g_kernel_variant[QV_KV_COUNT] = {{-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}};
void qv_kernel_variant_set(int which, int mode) {
    if (which >= 0 && which < QV_KV_COUNT) g_kernel_variant[which].store(mode < 0 ? -1 : mode);
}
int qv_kernel_variant(int which) {
    static const struct Env { int v[QV_KV_COUNT]; Env() {
        const char *names[QV_KV_COUNT] = {"QVERSE_LOGMEL", "QVERSE_ORT_SUB", "QVERSE_SPANS", "QVERSE_FWD_GRAPH", "QVERSE_CTC", nullptr, nullptr, nullptr};
        const int dflt[QV_KV_COUNT] = {1, 1, 1, 1, 1, 0, 0, 0};
        for (int i = 0; i < QV_KV_COUNT; ++i) {
            const char *e = names[i] ? getenv(names[i]) : nullptr;
            v[i] = (e && e[0] >= '0' && e[0] <= '9') ? atoi(e) : dflt[i];
        }
    } } env;
    if (which < 0 || which >= QV_KV_COUNT) return 0;
    const int o = g_kernel_variant[which].load();
    return o < 0 ? env.v[which] : o;
}

void launch_logmel(const float *audio, int64_t n_max, const int32_t *n_samples, const FrontendTab &ft, float *feats,
                   int tm_max, double *stats, int batch, hipStream_t s) {
    double *part = stats + (size_t)batch * 2 * QV_NMEL;
    if (qv_kernel_variant(QV_KV_LOGMEL) == 1)
        hipLaunchKernelGGL(lmv::k_logmel_reg<0>, dim3((tm_max + 3) / 4, batch), dim3(256), 0, s, audio, n_max, n_samples, ft, feats, tm_max);
    else
        hipLaunchKernelGGL(k_logmel, dim3((tm_max + 3) / 4, batch), dim3(256), 0, s, audio, n_max, n_samples, ft, feats, tm_max);
    hipLaunchKernelGGL(k_melstats, dim3(MS_CHUNKS, batch), dim3(320), 0, s, feats, n_samples, tm_max, part);
    hipLaunchKernelGGL(k_melstats_sum, dim3(batch), dim3(2 * QV_NMEL), 0, s, part, stats);
}

void launch_melapply(const float *feats, const int32_t *n_samples, int tm_max, const double *stats, float *out, int batch,
                     hipStream_t s) {
    hipLaunchKernelGGL(k_melapply, dim3(64, batch), dim3(320), 0, s, feats, n_samples, tm_max, stats, out);
}

void launch_conv0(const float *feats, int tm_max, const int32_t *len_in, const double *stats, const float *w,
                  const float *bias, half_t *out, int t1_max, int batch, hipStream_t s) {
    hipLaunchKernelGGL(k_conv0, dim3(1, t1_max, batch), dim3(256), 0, s, feats, tm_max, len_in, stats, w, bias, out, t1_max);
}

void launch_sub01(const float *feats, int tm_max, const int32_t *len_mel, const double *stats, const float *w0,
                  const float *b0, const int32_t *len1, const float *w1, const float *b1, half_t *out, int t2_max, int batch,
                  hipStream_t s) {
    hipLaunchKernelGGL(k_sub01, dim3(1, (t2_max + SUB_TT - 1) / SUB_TT, batch), dim3(256), 0, s, feats, tm_max,
                       len_mel, stats, w0, b0, len1, w1, b1, out, t2_max);
}

void launch_dwconv2d(const half_t *in, int tin_max, int fin, const int32_t *len_in, const float *w, const float *bias,
                     half_t *out, int tout_max, int fout, int batch, hipStream_t s) {
    hipLaunchKernelGGL(k_dwconv2d, dim3(1, (tout_max + DW2_TT - 1) / DW2_TT, batch), dim3(256), 0, s, in, tin_max, fin, len_in, w,
                       bias, out, tout_max, fout);
}

void launch_pack_rows(const half_t *x, int t_max, int row_elems, const int32_t *len, const int32_t *row_off, half_t *y,
                      int32_t *row_map, int batch, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_rows, dim3(1, t_max, batch), dim3(256), 0, s, x, t_max, row_elems, len, row_off, y, row_map);
}

void launch_layernorm(const float *x, const float *g, const float *b, half_t *y, int M, hipStream_t s) {
    hipLaunchKernelGGL(k_layernorm, dim3((M + 4 * LN_ROWS - 1) / (4 * LN_ROWS)), dim3(256), 0, s, x, g, b, y, M);
}

void launch_layernorm2(float *x, const float *g1, const float *b1, const float *g2, const float *b2, half_t *y, int M,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_layernorm2, dim3((M + 4 * LN_ROWS - 1) / (4 * LN_ROWS)), dim3(256), 0, s, x, g1, b1, g2, b2, y, M);
}

void launch_to_half(const float *x, half_t *y, size_t n, hipStream_t s) {
    size_t n8 = n / 8;
    hipLaunchKernelGGL(k_to_half, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, x, y, n8);
}

void launch_to_float(const half_t *x, float *y, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_to_float, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
}

// Variants of the attention kernel (tests/test_gpu_forward.py): 3 = the default (an utterance of at most ATT_SHORT_T
// frames on k_attention_short, a longer one on k_attention_ws with two heads per block), 0 = k_attention_ws with two heads per
// block for every utterance, 1 = one head per block (3 K/V stages, 256-row ring), 2 = the one-wave-per-query-tile kernel the
// specialised ones replaced, 4 = k_attention_x (round 6: four self-staging waves per (head, query group), two blocks per CU)
// for every utterance, 5 = k_attention_short + k_attention_x; 0, 1, 2 and 4 are bit-identical, so are 3 and 5.
// k_attention_x is 4-10 % faster than k_attention_ws launched back to back (tools/att_bench) and level with it inside the
// forward with batches in flight (profiles/r06_p_attention_x_same_box.log), so it stays a variant.  The environment
// (QVERSE_ATT_X=1 / QVERSE_ATT_HPB=1 / QVERSE_ATT_OLD=1 / QVERSE_ATT_TILED=1) is read ONCE per process; tests switch with
// qv_debug_attention_variant() instead of setenv, which is not safe against launches from another thread.
static std::atomic<int> g_att_variant{-1};
void qv_attention_set_variant(int mode) { g_att_variant.store(mode); }
static int attention_variant() {
    static const int env = [] {
        // QVERSE_ATT_OLD=1: one wave per query tile; QVERSE_ATT_HPB=1: one head per block; QVERSE_ATT_TILED=1: the
        // two-heads-per-block kernel for every utterance (no k_attention_short)
        // ... QVERSE_ATT_X=1: k_attention_x instead of k_attention_ws for the long utterances (variant 5)
        const char *o = getenv("QVERSE_ATT_OLD"), *h = getenv("QVERSE_ATT_HPB"), *t = getenv("QVERSE_ATT_TILED"), *x = getenv("QVERSE_ATT_X");
        return (o && o[0] == '1') ? 2 : (h && h[0] == '1') ? 1 : (t && t[0] == '1') ? 0 : (x && x[0] == '1') ? 5 : 3;
    }();
    const int v = g_att_variant.load();
    return v < 0 ? env : v;
}

int qv_attention_variant() { return attention_variant(); }

void launch_attention(const half_t *qk, const half_t *vt, const half_t *pos, int pos_ld, const float *bu, const float *bv,
                      const int32_t *len, const int32_t *row_off, half_t *out, int t_max, int t_min, int t_pad, int batch,
                      hipStream_t s, int variant_in) {
    // a forward pass passes the variant it read ONCE (qv_model.hip): a test that flips the process-wide switch while
    // batches of other contexts are in flight cannot change kernels between the layers of one forward
    const int variant = variant_in >= 0 ? variant_in : attention_variant();
    if (variant == 2) {
        hipLaunchKernelGGL(k_attention, dim3(QV_H, batch), dim3(256), 0, s, qk, vt, pos, pos_ld, bu, bv, len, row_off, out, t_max,
                           t_pad);
        return;
    }
    // default (3): an utterance of at most ATT_SHORT_T frames goes to k_attention_short, a longer one to the
    // wave-specialised kernel -- by its OWN length, so that its bits do not depend on the batch it travels in; a launch
    // none of whose utterances qualify is skipped (t_min / t_max are the batch's shortest / longest utterance)
    int t_short = 0;
    if (variant == 3 || variant == 5) {      // 5: k_attention_short for short utterances, k_attention_x for the longer ones
        t_short = ATT_SHORT_T;
        if (t_min <= ATT_SHORT_T)
            hipLaunchKernelGGL(k_attention_short, dim3(QV_H, batch), dim3(256), 0, s, qk, vt, pos, pos_ld, bu, bv, len, row_off,
                               out, t_max, t_pad);
        if (t_max <= ATT_SHORT_T) return;
    }
    if (variant == 4 || variant == 5)
        hipLaunchKernelGGL((k_attention_x<192>), dim3(QV_H, (t_max + 127) / 128, batch), dim3(256), 0, s, qk, vt, pos,
                           pos_ld, bu, bv, len, row_off, out, t_max, t_pad, t_short);
    else if (variant == 1)
        hipLaunchKernelGGL((k_attention_ws<1, 3, 256>), dim3(QV_H, (t_max + 127) / 128, batch), dim3(512), 0, s, qk, vt, pos,
                           pos_ld, bu, bv, len, row_off, out, t_max, t_pad, t_short);
    else
        hipLaunchKernelGGL((k_attention_ws<2, 2, 192>), dim3(QV_H / 2, (t_max + 127) / 128, batch), dim3(768), 0, s, qk, vt, pos,
                           pos_ld, bu, bv, len, row_off, out, t_max, t_pad, t_short);
}

void launch_dwconv1d(const half_t *x, const float *w, const float *bias, const int32_t *len, const int32_t *row_off, half_t *y,
                     int t_max, int batch, hipStream_t s) {
    hipLaunchKernelGGL(k_dwconv1d, dim3((t_max + 4 * DW_TT - 1) / (4 * DW_TT), batch), dim3(256), 0, s, x, w, bias, len, row_off,
                       y);
}

void launch_logsoftmax(const float *logits, int ld, float *out, int M, const int32_t *row_map, int t_out, hipStream_t s) {
    hipLaunchKernelGGL(k_logsoftmax, dim3((M + 3) / 4), dim3(256), 0, s, logits, ld, out, M, row_map, t_out);
}

void launch_zero_pad_rows(float *out, const int32_t *len, int t_out, int t_min, int batch, hipStream_t s) {
    if (t_out <= t_min) return;   // every utterance fills the tensor
    hipLaunchKernelGGL(k_zero_pad_rows, dim3(t_out - t_min, batch), dim3(256), 0, s, out, len, t_out);
}

// ---------------------------------------------------------------- a15: polyphase resampler ----
// scipy.signal.upfirdn's float32 inner loop, one output sample per thread (tta/run.py:60-71 reaches
// it through resample_poly).  hflip[t][j] = h[t + up * (P - 1 - j)] (zero beyond n_taps) is built
// on the host, so a thread walks x forwards and its phase row forwards, exactly scipy's order;
// out-of-range inputs are skipped like scipy's index clamping does, not added as zeros.
namespace {
__global__ __launch_bounds__(256) void k_upfirdn(const float *__restrict__ x, int64_t n_in, const float *__restrict__ hflip,
                                                 int P, int up, int down, int64_t m0, int64_t n_out, float *__restrict__ y) {
    int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= n_out) return;
    int64_t pos = (m0 + m) * down;
    int64_t xi = pos / up;
    int t = (int)(pos - xi * up);
    const float *h = hflip + (size_t)t * P;
    int64_t first = xi - (P - 1);
    int j0 = first < 0 ? (int)(-first) : 0;
    int64_t last = xi < n_in - 1 ? xi : n_in - 1;   // last valid input index
    float acc = 0.f;
    for (int64_t i = first + j0; i <= last; ++i) acc = __fadd_rn(acc, __fmul_rn(x[i], h[i - first]));
    y[m] = acc;
}
// The same FIR for a BATCH of rows in one launch (round 6: the TTA wrapper's 0.9x / 1.1x copies of every gated clip, the
// 44.1 / 48 kHz -> 16 kHz ingest of audio.load_audio_device): output row r reads source row src[r] (or r) of x, n_in[r]
// samples long, writes ceil(n_in[r] * up / down) samples and zeroes the rest of its pitch -- the engine's [B, N]
// zero-padded input layout.  Per output sample the arithmetic is k_upfirdn's, term for term.
__global__ __launch_bounds__(256) void k_upfirdn_rows_direct(const float *__restrict__ x, int64_t x_pitch, const int32_t *__restrict__ src,
                                                             const int64_t *__restrict__ n_in_rows, const float *__restrict__ hflip,
                                                             int P, int up, int down, int64_t m0, float *__restrict__ y, int64_t y_pitch) {
    const int r = blockIdx.y;
    const int64_t n_in = n_in_rows[r];
    const int64_t n_out = (n_in * up + down - 1) / down;
    const float *xr = x + (size_t)(src ? src[r] : r) * x_pitch;
    float *yr = y + (size_t)r * y_pitch;
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < y_pitch; m += (int64_t)gridDim.x * 256) {
        if (m >= n_out) { yr[m] = 0.f; continue; }
        int64_t pos = (m0 + m) * down;
        int64_t xi = pos / up;
        int t = (int)(pos - xi * up);
        const float *h = hflip + (size_t)t * P;
        int64_t first = xi - (P - 1);
        int j0 = first < 0 ? (int)(-first) : 0;
        int64_t last = xi < n_in - 1 ? xi : n_in - 1;
        float acc = 0.f;
        for (int64_t i = first + j0; i <= last; ++i) acc = __fadd_rn(acc, __fmul_rn(xr[i], h[i - first]));
        yr[m] = acc;
    }
}

// ... and with the operands staged in LDS: the 256 outputs of a block read one window of x (256 * down / up + P samples) and
// the whole per-phase filter (up * P taps); from global memory every thread walked its P taps with two dependent loads per
// product (809 us per launch for 64 clips x 33 s in the TTA profile of round 6).  Same products, same order, same bits.
__global__ __launch_bounds__(256) void k_upfirdn_rows(const float *__restrict__ x, int64_t x_pitch, const int32_t *__restrict__ src,
                                                      const int64_t *__restrict__ n_in_rows, const float *__restrict__ hflip,
                                                      int P, int up, int down, int64_t m0, float *__restrict__ y, int64_t y_pitch, int xspan) {
    extern __shared__ __attribute__((aligned(16))) float sm_up[];
    float *sh = sm_up, *sx = sm_up + up * P;
    const int r = blockIdx.y, tid = threadIdx.x;
    const int64_t n_in = n_in_rows[r];
    const int64_t n_out = (n_in * up + down - 1) / down;
    const float *xr = x + (size_t)(src ? src[r] : r) * x_pitch;
    float *yr = y + (size_t)r * y_pitch;
    for (int i = tid; i < up * P; i += 256) sh[i] = hflip[i];
    for (int64_t mb = (int64_t)blockIdx.x * 256; mb < y_pitch; mb += (int64_t)gridDim.x * 256) {
        const int64_t m = mb + tid;
        if (mb >= n_out) {                                   // (uniform: the whole block is in the zero padding)
            if (m < y_pitch) yr[m] = 0.f;
            continue;
        }
        const int64_t m_hi = mb + 255 < n_out - 1 ? mb + 255 : n_out - 1;
        int64_t x_lo = ((m0 + mb) * down) / up - (P - 1), x_hi = ((m0 + m_hi) * down) / up;
        x_lo = x_lo < 0 ? 0 : x_lo;
        x_hi = x_hi < n_in - 1 ? x_hi : n_in - 1;
        __syncthreads();                                     // the previous window has been consumed (and the taps are in place)
        for (int64_t i = x_lo + tid; i <= x_hi; i += 256) sx[i - x_lo] = xr[i];
        __syncthreads();
        if (m >= y_pitch) continue;
        if (m >= n_out) { yr[m] = 0.f; continue; }
        const int64_t pos = (m0 + m) * down;
        const int64_t xi = pos / up;
        const int t = (int)(pos - xi * up);
        const float *h = sh + t * P;
        const int64_t first = xi - (P - 1);
        const int j0 = first < 0 ? (int)(-first) : 0;
        const int64_t last = xi < n_in - 1 ? xi : n_in - 1;
        float acc = 0.f;
        for (int64_t i = first + j0; i <= last; ++i) acc = __fadd_rn(acc, __fmul_rn(sx[i - x_lo], h[i - first]));
        yr[m] = acc;
    }
}

// interleaved [frames][channels] float32 -> mono: numpy's float32 mean over the channel axis (sequential sum, one
// division by the channel count), what audio._read_wav / the reference's soundfile fallback do (shared/audio.py:13-15)
__global__ __launch_bounds__(256) void k_mixdown(const float *__restrict__ x, int64_t x_pitch, const int64_t *__restrict__ n_frames,
                                                 int channels, float *__restrict__ y, int64_t y_pitch) {
    const int r = blockIdx.y;
    const int64_t n = n_frames[r];
    const float *xr = x + (size_t)r * x_pitch;
    float *yr = y + (size_t)r * y_pitch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float acc = xr[i * channels];
        for (int c = 1; c < channels; ++c) acc = __fadd_rn(acc, xr[i * channels + c]);
        yr[i] = __fdiv_rn(acc, (float)channels);
    }
}
}  // namespace

void launch_upfirdn_rows(const float *x, int64_t x_pitch, const int32_t *src, const int64_t *n_in_rows, int rows, const float *hflip,
                         int P, int up, int down, int64_t m0, float *y, int64_t y_pitch, hipStream_t s) {
    if (rows <= 0 || y_pitch <= 0) return;
    const int64_t bx = (y_pitch + 255) / 256;
    // window of x a block of 256 outputs reads (+ 2: the two floor divisions at its ends) and the filter, in LDS when they fit
    const int64_t xspan = (256 * (int64_t)down) / up + P + 2;
    const size_t lds = ((size_t)up * P + (size_t)xspan) * sizeof(float);
    if (lds <= 60 * 1024)
        hipLaunchKernelGGL(k_upfirdn_rows, dim3((unsigned)(bx < 4096 ? bx : 4096), rows), dim3(256), lds, s, x, x_pitch, src, n_in_rows, hflip, P, up,
                           down, m0, y, y_pitch, (int)xspan);
    else
        hipLaunchKernelGGL(k_upfirdn_rows_direct, dim3((unsigned)(bx < 4096 ? bx : 4096), rows), dim3(256), 0, s, x, x_pitch, src, n_in_rows, hflip,
                           P, up, down, m0, y, y_pitch);
}

void launch_mixdown(const float *x, int64_t x_pitch, const int64_t *n_frames, int rows, int channels, float *y, int64_t y_pitch,
                    int64_t max_frames, hipStream_t s) {
    if (rows <= 0 || max_frames <= 0) return;
    const int64_t bx = (max_frames + 255) / 256;
    hipLaunchKernelGGL(k_mixdown, dim3((unsigned)(bx < 4096 ? bx : 4096), rows), dim3(256), 0, s, x, x_pitch, n_frames, channels, y, y_pitch);
}

void launch_upfirdn(const float *x, int64_t n_in, const float *hflip, int P, int up, int down, int64_t m0, int64_t n_out,
                    float *y, hipStream_t s) {
    if (n_out <= 0) return;
    hipLaunchKernelGGL(k_upfirdn, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, x, n_in, hflip, P, up, down, m0,
                       n_out, y);
}
