// qv_gemm_dequant.h -- shared by qv_gemm.hip and qv_gemm256.hip: int4 / int8 weight codes -> f16 MFMA operands, and the
// packed-f32 Swish / sigmoid of the epilogues.
#pragma once

#include "qv_kernels.h"

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// 8 int4 codes (one 4-byte chunk of the W4 layout) -> 8 halves (q - zp) * scale, in k order.
// 0x6400 | x is the half 1024 + x, so the integer->float conversion is one OR; the subtraction of
// off = 1024 + zp is exact and the product rounds once, i.e. the result is half((q - zp) * scale).
static __device__ __forceinline__ half8 dequant8(uint32_t q, half2_t s2, half2_t off) {
    half8 r;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        uint32_t bits = ((q >> (4 * p)) & 0x000F000Fu) | 0x64006400u;
        half2_t h = (__builtin_bit_cast(half2_t, bits) - off) * s2;
        r[2 * p] = h[0];
        r[2 * p + 1] = h[1];
    }
    return r;
}

// 8 bytes u = q + 128 -> 8 halves q, exactly: a byte next to 0x64 is the half 1024 + u (v_perm_b32 puts
// it there), and (1024 + u) - 1152 = q needs no rounding.  The per-channel scale is applied in the epilogue.
static __device__ __forceinline__ half8 dequant8_i8(uint2 qv) {
    const half2_t off = {(_Float16)1152.0f, (_Float16)1152.0f};
    half8 r;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t src = p < 2 ? qv.x : qv.y;
        const uint32_t bits = __builtin_amdgcn_perm(0x64646464u, src, (p & 1) ? 0x04030402u : 0x04010400u);
        const half2_t h = __builtin_bit_cast(half2_t, bits) - off;
        r[2 * p] = h[0];
        r[2 * p + 1] = h[1];
    }
    return r;
}

// x * sigmoid(x) for four values, the scalings and the + 1 as packed f32 operations (v_pk_mul_f32 / v_pk_add_f32
// handle two values per instruction; the transcendentals stay scalar).  Same values as x * sigm(x): __expf(-x) is
// exp2(-x * log2 e), and (-x) * c == -(x * c) exactly.
typedef float f32x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ f32x4 sigmoid4(f32x4 x) {
    const f32x2 L2E = {0x1.715476p+0f, 0x1.715476p+0f}, ONE = {1.0f, 1.0f};
    f32x4 r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 xv = {x[2 * h], x[2 * h + 1]};
        const f32x2 y = xv * L2E;
        const f32x2 e = {__builtin_amdgcn_exp2f(-y[0]), __builtin_amdgcn_exp2f(-y[1])};
        const f32x2 d = e + ONE;
        r[2 * h] = __builtin_amdgcn_rcpf(d[0]);
        r[2 * h + 1] = __builtin_amdgcn_rcpf(d[1]);
    }
    return r;
}
static __device__ __forceinline__ f32x4 swish4(f32x4 x) {
    const f32x2 L2E = {0x1.715476p+0f, 0x1.715476p+0f}, ONE = {1.0f, 1.0f};
    f32x4 r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 xv = {x[2 * h], x[2 * h + 1]};
        const f32x2 y = xv * L2E;
        const f32x2 e = {__builtin_amdgcn_exp2f(-y[0]), __builtin_amdgcn_exp2f(-y[1])};
        const f32x2 d = e + ONE;
        const f32x2 s = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        const f32x2 o = xv * s;
        r[2 * h] = o[0];
        r[2 * h + 1] = o[1];
    }
    return r;
}
