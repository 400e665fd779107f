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
#include "qv_gemm_body.inc"
}

// Round 6 (advisor): the whole library is built WITHOUT packed-FP32 code generation (offline-tarteel_amd/build.py: the
// v_pk_*_f32 class is what the cross-kernel disturbance of DESIGN.md 4.1 corrupts) -- the GEMM translation units included.
// Two instantiations of the 128-wide kernel (W8A16, register-staged loaders, two stages: GLU and residual epilogue, the
// pointwise convolutions of precision 1) do not fit the 128-VGPR budget of their 512-thread block without the packed forms
// (7 spilled registers, and a spill's scratch traffic would break the loaders' hand-counted vmcnt).  They alone keep the
// feature, as their own kernel symbol: tests/test_capi_load.py allows v_pk_*_f32 in k_gemm_pk<...> and nowhere else, and
// the interference probe runs them as victims (tests/test_gpu_interference.py).
template <int EPI, int BN, int WQ, int NST, int LD>
__global__ __launch_bounds__(512, 4) __attribute__((target("packed-fp32-ops"))) void k_gemm_pk(GemmArgs g) {
#include "qv_gemm_body.inc"
}

template <int EPI, int BN, int WQ, int NST, int LD>
constexpr bool gemm_needs_pk() { return WQ == 8 && BN == 128 && NST == 2 && LD == 1 && (EPI == EPI_GLU || EPI == EPI_RESID); }

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
            if constexpr (gemm_needs_pk<EPI, BN, WQ, NST, LD>())
                (void)hipFuncSetAttribute((const void *)k_gemm_pk<EPI, BN, WQ, NST, LD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            else
                (void)hipFuncSetAttribute((const void *)k_gemm<EPI, BN, WQ, NST, LD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            allowed = lds;
        }
    }
    if constexpr (gemm_needs_pk<EPI, BN, WQ, NST, LD>()) hipLaunchKernelGGL((k_gemm_pk<EPI, BN, WQ, NST, LD>), grid, dim3(512), lds, s, g);
    else hipLaunchKernelGGL((k_gemm<EPI, BN, WQ, NST, LD>), grid, dim3(512), lds, s, g);
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
