// qv_kernels.h -- device kernels of the acoustic model (FastConformer-CTC forward), gfx950.
//
// Layout conventions (row = one encoder frame of one utterance, M = B * t_max rows):
//   residual stream   f32 [M][512]
//   GEMM operands     f16, A [M][K] row-major, W [N][K] row-major ("NT": both K-contiguous, the
//                     natural MFMA fragment order: a lane reads 8 consecutive k of one row)
//   accumulation      f32 in MFMA accumulators; bias / activation / residual fused in the epilogue
//
// GEMM: 128 x BN x 64 tiles, 512 threads (4 consumer waves as 2 x 2 on v_mfma_f32_32x32x16_f16 +
// 4 loader waves), operands staged with direct global->LDS loads (16 B per lane).  LDS rows are
// 128 B; the 16-B chunk index is XOR-swizzled with (row >> 1) & 7 ON THE GLOBAL SOURCE ADDRESS (the
// LDS write of a direct load is lane-linear) and on the fragment read, which makes the
// ds_read_b128 fragment reads conflict-free.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define QV_D 512
#define QV_H 8
#define QV_DK 64
#define QV_FF 2048
#define QV_NMEL 80
#define QV_SUBC 256

enum {
    EPI_F16 = 0,        // out f16 = acc + bias
    EPI_F16_SWISH = 1,  // out f16 = swish(acc + bias)
    EPI_F16_RELU = 2,   // out f16 = relu(acc + bias)
    EPI_GLU = 3,        // out f16 [M][N/2]: (acc_a + b_a) * sigmoid(acc_g + b_g), W rows pre-interleaved
    EPI_RESID = 4,      // out f32 [M][N] = out + alpha * (acc + bias)   (in place on the residual stream)
    EPI_F32 = 5,        // out f32 = alpha * (acc + bias)
    EPI_QKV = 6,        // N = 1536: q,k -> f16 [M][1024] (+bias), v -> transposed f16 Vt[b][h][64][t_pad]
    EPI_F32_RELU = 7    // out f32 = relu(acc + bias)   (A8W8 only: the ReLU output feeds the next quantiser)
};

struct GemmArgs {
    const half_t *A;      // [M][lda]
    const half_t *W;      // [N][ldw]
    const float *bias;    // [N] or nullptr
    void *out;
    void *out2;           // EPI_QKV: Vt
    int M, N, K, lda, ldw, ldo;
    float alpha;
    int in_flight;        // batches the engine keeps in flight on the device (execution contexts); 0/1 = one -- tile policy only
    int t_max, t_pad;     // EPI_QKV
    const int32_t *row_map;  // EPI_QKV: [M] (utterance << 16 | frame) of each packed row
    // W4A16 variant (Wq != nullptr, W unused): block-128 int4 weights, w = (q - zero_point) * scale.
    //   Wq      [N/64][K/64][64 rows][32 B]: one 64 x 64 tile of nibbles is 2 KB contiguous; inside a
    //           row's 32 B the 4-byte chunk c (k = 8c .. 8c+7) sits at position c ^ ((n >> 2) & 7),
    //           and in a chunk nibble p holds k = 2p, nibble p + 4 holds k = 2p + 1 (p = 0..3)
    //   wscale  f16 [K/128][N][2] = {scale, 1024 + zero_point}
    const uint8_t *Wq;
    const half_t *wscale;
    // W8A16 variant (W8 != nullptr; the pointwise convolutions under QV_PREC_MIXED_INT4_INT8): per-
    // output-channel symmetric int8, w = q * w8scale[n].
    //   W8      [N/64][K/64][64 rows][64 B]: one 64 x 64 tile is 4 KB contiguous; a byte holds q + 128;
    //           inside a row the 8-byte chunk c (k = 8c .. 8c+7) sits at position c ^ ((n >> 2) & 7)
    //   w8scale f32 [N], applied in the epilogue (the MFMA runs on the exact integers)
    const uint8_t *W8;
    const float *w8scale;
    // A8W8 variant (Wi8 != nullptr; QV_PREC_ORT_MIXED, qv_ort.h): A is s8 [M][2 * lda] = x_q - 128, Wi8 is s8 [N][2 * ldw]
    // row-major (K, lda, ldw stay in units of 2 bytes, so the staging code is the f16 kernel's byte for byte), the
    // products run on v_mfma_i32_32x32x32_i8 and the epilogue turns the int32 accumulator into
    //   float(acc + (128 - zp_x[u]) * wsum[n]) * (s_x[u] * w_scale) + bias[n],  u = the row's utterance,
    // with {s_x, zp_x} from the {min, max} keys mm_in[2u .. 2u+1].  mm_out (GLU / ReLU epilogues): the output range of
    // the valid rows is folded into mm_out[2u .. 2u+1] for the next quantiser.
    const int8_t *Wi8;
    const int32_t *wsum;      // [N] sum over k of Wi8[n][k]
    float w_scale;
    const uint32_t *mm_in;
    uint32_t *mm_out;
    const int32_t *len;       // dense rows (row_map == nullptr): valid frames per utterance
    int rows_per_utt, f_per_t;
#ifdef QV_GEMM_TRACE
    // dev tool only (tools/gemm_trace.hip): [block][wave][K-step][4] s_memtime stamps
    unsigned long long *trace;
    unsigned long long *phase;   // [block][4] s_memrealtime (100 MHz) at entry, first barrier release, K loop end, exit
    int abl;   // ablation mask: 1 no MFMA, 2 no fragment reads, 4 no ds_write (LD = 1), 8 no loads (LD = 1)
#endif
};

// (dev switch for A/B builds of tools/gemm_bench.hip: -DQV_GEMM_NOSWAP = stage stores first, then the re-requests)
#ifdef QV_GEMM_NOSWAP
#define QV_SWAP false
#else
#define QV_SWAP true
#endif

// WQ: 0 = f16 weights, 4 = W4A16, 8 = W8A16, 88 = A8W8 (int8 x int8 -> int32)
// LD: 0 = direct global->LDS loads with NST LDS stages, 1 = register-staged loader waves (NST = 2)
template <int EPI, int BN, int WQ, int NST, int LD>
__global__ void k_gemm(GemmArgs g);

// host-side packer for the W4A16 layout above (w: [N][K] f32 row-major; N % 64 == 0, K % 128 == 0)
void qv_pack_w4(const float *w, int N, int K, uint8_t *q_out, half_t *scale_out);
// ... from a GIVEN grid (a pre-quantised weight file's own MatMulNBits blocks): scale, zp [N][K/128]; returns the count of
// elements that did not sit on that grid (0 for a consistent file)
int64_t qv_pack_w4_given(const float *w, int N, int K, const float *scale, const float *zp, uint8_t *q_out, half_t *scale_out);
// ... and for the W8A16 layout (N % 64 == 0, K % 64 == 0): scale = max|w| / 127 per row, q = round(w / scale)
void qv_pack_w8(const float *w, int N, int K, uint8_t *q_out, float *scale_out);

void launch_gemm(int epi, const GemmArgs &g, hipStream_t s);
// 256 x 256 tiles, one block per CU (qv_gemm256.hip): N % 256 == 0 only; false = nothing launched
bool launch_gemm256(int epi, const GemmArgs &g, hipStream_t s, int bm = 256);   // bm: tile height, 256 or 192
// tile policy of launch_gemm: 0 = 128-wide tiles only, 1 = 256 x 256 where the grid has >= 160 tiles (default), 2 = 256 x 256 wherever the shape allows; -1 = back to QVERSE_GEMM_T256 / the default
void qv_gemm_set_t256(int mode);
// tile height of the wide kernel: 0 = 256 rows always, 1 = 192 rows where that saves a round of tiles and fewer than three batches
// are in flight (default), 2 = ... whatever is in flight, 3 = 192 rows always; -1 = back to QVERSE_GEMM_BM / the default
void qv_gemm_set_bm(int mode);
// changes whenever a qv_gemm_set_* switch does: part of the forward-graph key (a captured graph holds the old kernel choice)
int qv_gemm_policy_epoch();
// tools/gemm_bench only (QV_GEMM_Q_VARIANT builds): 256 x 256 tiles with four waves of 128 x 128 (tools/gemm256q.h)
void qv_gemm_set_q(int mode);
// the kernel launch_gemm picks for this call, e.g. "k_gemm256<f16_swish>" / "k_gemm<resid,128>" (thread-local buffer)
const char *qv_gemm_kernel_name(int epi, const GemmArgs &g);

// Cross-check variants of single kernels (qv_debug_kernel_variant): the override if one is set, else the environment
// variable (read once per process), else the default.  Variants of one kernel give identical bits unless stated.
//   QV_KV_LOGMEL  (QVERSE_LOGMEL)   1 = FFT in registers, cross-lane strides over DPP / v_permlane*_swap (default);
//                                   0 = Stockham FFT through LDS (the kernel of rounds 1-4)
//   QV_KV_ORT_SUB (QVERSE_ORT_SUB)  precision 2's conv.0: 1 = f32 matrix pipe, four channel groups per block (default); 0 = VALU
//   QV_KV_SPANS   (QVERSE_SPANS)    match_verse's span pass: 1 = prefix-shared walk per start verse (k_spans2, default); 0 = one walk per span
//   QV_KV_FWD_GRAPH (QVERSE_FWD_GRAPH) multi-context engines: 1 = a forward whose shape repeats on a context is replayed as one hipGraph
//                                   launch (default); 0 = always the plain launches
//   QV_KV_CTC     (QVERSE_CTC)      the alpha recursion of the CTC rerank: 1 = parity-specialised wave program (ctc_wave2, default);
//                                   0 = the wave program of rounds 1-5
enum { QV_KV_LOGMEL = 0, QV_KV_ORT_SUB = 1, QV_KV_SPANS = 2, QV_KV_FWD_GRAPH = 3, QV_KV_CTC = 4, QV_KV_COUNT = 8 };
int qv_kernel_variant(int which);
void qv_kernel_variant_set(int which, int mode);   // mode < 0: back to the environment / default

// measurement hooks (bench.py roofline): per-launch HIP-event timing of the GEMMs
void qv_gemm_prof_enable(bool on);
bool qv_gemm_prof_on();   // per-launch event timing is active: the forward must issue real launches (no graph replay)
void qv_gemm_prof_collect(double *ms, double *flops, int *n);
