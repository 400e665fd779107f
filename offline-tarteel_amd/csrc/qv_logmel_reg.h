// qv_logmel_reg.h -- the log-mel kernel with its 256-point FFT in registers (round 5).
//
// Same transform, same butterflies on the same operands in the same order as k_logmel (qv_layers.hip: radix-2 Stockham,
// N = 256 on the packed real frame), hence the same bits -- only the data movement differs: a lane holds four complex
// points, the two in-lane strides are local butterflies and the six cross-lane strides are exchanges over DPP
// (quad_perm, row_half_mirror, row_ror) and v_permlane16/32_swap, i.e. VALU instructions; LDS memory is not touched before
// the power spectrum.  Why: a co-running kernel of another engine (f16 MFMA fed from LDS at full rate, tools/withdrawn/)
// was seen to disturb the LDS-exchanging kernel (tools/interference_probe.hip); this one has no LDS
// exchange to disturb, and it is VALU-bound instead of LDS-latency-bound.
// Included by qv_layers.hip (the product kernel) and by tools/logmel_variants.h (the probe's victim 7).
#pragma once

#include "qv_layers.h"

namespace lmv {

__device__ __forceinline__ float2 cmul_tw(float2 v, float2 w) {      // the shipped kernel's v * w, operation for operation
    return make_float2(__builtin_fmaf(v.x, w.x, -(v.y * w.y)), __builtin_fmaf(v.x, w.y, v.y * w.x));
}

// value of lane (lane ^ (1 << Q)); VALU only.  DPP controls: quad_perm 0x00-0xFF, row_ror:n 0x120 + n, row_mirror 0x140,
// row_half_mirror 0x141.  The two swap instructions exchange halves of a register PAIR; fed the same value twice, one of
// the two results is the lane's own value and the other one the partner's -- picked by comparing bits, so the code does
// not depend on which operand receives which half.
template <int Q> __device__ __forceinline__ float lane_xor(float v) {
    const int x = __float_as_int(v);
    int r;
    if (Q == 0) r = __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true);            // quad_perm [1,0,3,2]
    else if (Q == 1) r = __builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true);       // quad_perm [2,3,0,1]
    else if (Q == 2) {                                                            // i ^ 7 then i ^ 3
        r = __builtin_amdgcn_mov_dpp(x, 0x141, 0xf, 0xf, true);
        r = __builtin_amdgcn_mov_dpp(r, 0x1B, 0xf, 0xf, true);
    } else if (Q == 3) r = __builtin_amdgcn_mov_dpp(x, 0x128, 0xf, 0xf, true);    // row_ror:8
    else if (Q == 4) {
        const auto p = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
        r = (int)(p[0] == (unsigned)x ? p[1] : p[0]);
    } else {
        const auto p = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
        r = (int)(p[0] == (unsigned)x ? p[1] : p[0]);
    }
    return __int_as_float(r);
}

// one cross-lane radix-2 pass over lane bit Q: the lane whose bit is clear holds u, its partner v; u' = u + v w, v' = u - v w
template <int Q> __device__ __forceinline__ void cross_pass(float2 z[4], float2 w, int lane) {
    const bool hi = (lane >> Q) & 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float2 vw = cmul_tw(z[r], w);
        const float2 s = hi ? vw : z[r];
        const float2 o = make_float2(lane_xor<Q>(s.x), lane_xor<Q>(s.y));
        z[r] = hi ? make_float2(o.x - s.x, o.y - s.y) : make_float2(s.x + o.x, s.y + o.y);
    }
}

// MIRROR 0: ds_bpermute for the k <-> 256 - k exchange.  Everything before the power spectrum stays in registers.
template <int MIRROR>
__global__ __launch_bounds__(256) void k_logmel_reg(const float *__restrict__ audio, int64_t n_max,
                                                    const int32_t *__restrict__ n_samples, const FrontendTab ft,
                                                    float *__restrict__ feats, int tm_max) {
    __shared__ float pw[4][264];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.y, t = blockIdx.x * 4 + wave;
    const int n = n_samples[b];
    const int tm = n / 160 + 1;
    if (t >= tm) return;
    const float *x = audio + (size_t)b * n_max;
    // lane L holds z[n], n = 4 * bitrev6(L) + r: pass q (q = 0 .. 5) pairs the lanes that differ in bit q, passes 6 and 7
    // pair registers; after pass q bit q of the element's index is lane bit q, so the result is Z[L + 64 RA + 128 RB].
    const int rl = (int)(__builtin_bitreverse32((unsigned)lane) >> 26);
    float2 wq[6], wu[4];
#pragma unroll
    for (int q = 0; q < 6; ++q) wq[q] = ft.twiddle[(lane & ((1 << q) - 1)) * (256 >> q)];
    const float2 w6 = ft.twiddle[4 * lane], w7a = ft.twiddle[2 * lane], w7b = ft.twiddle[2 * lane + 128];
#pragma unroll
    for (int m = 0; m < 4; ++m) wu[m] = ft.twiddle[lane + 64 * m];
    float smp[8];
    {
        const int i0 = 8 * rl, s0 = t * 160 - 256 + i0;
        float win[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) win[j] = ft.window[i0 + j];
        if (t * 160 - 256 >= 1 && t * 160 + 255 < n) {        // wave-uniform: no reflection, no first sample
            float c[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) c[j] = x[s0 - 1 + j];
#pragma unroll
            for (int j = 0; j < 8; ++j) smp[j] = (c[j + 1] - 0.97f * c[j]) * win[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int s = s0 + j;
                if (s < 0) s = -s;
                if (s >= n) s = 2 * (n - 1) - s;
                s = s < 0 ? 0 : s;
                const float y = x[s] - (s > 0 ? 0.97f * x[s - 1] : 0.f);
                smp[j] = y * win[j];
            }
        }
    }
    float2 z[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = make_float2(smp[2 * r], smp[2 * r + 1]);
    cross_pass<0>(z, wq[0], lane);
    cross_pass<1>(z, wq[1], lane);
    cross_pass<2>(z, wq[2], lane);
    cross_pass<3>(z, wq[3], lane);
    cross_pass<4>(z, wq[4], lane);
    cross_pass<5>(z, wq[5], lane);
    {   // pass 6: register bit RA (z[0], z[2]) and (z[1], z[3]); k = lane
        const float2 a = cmul_tw(z[2], w6), c = cmul_tw(z[3], w6);
        const float2 u0 = z[0], u1 = z[1];
        z[0] = make_float2(u0.x + a.x, u0.y + a.y); z[2] = make_float2(u0.x - a.x, u0.y - a.y);
        z[1] = make_float2(u1.x + c.x, u1.y + c.y); z[3] = make_float2(u1.x - c.x, u1.y - c.y);
    }
    {   // pass 7: register bit RB (z[0], z[1]) with k = lane, (z[2], z[3]) with k = lane + 64
        const float2 a = cmul_tw(z[1], w7a), c = cmul_tw(z[3], w7b);
        const float2 u0 = z[0], u2 = z[2];
        z[0] = make_float2(u0.x + a.x, u0.y + a.y); z[1] = make_float2(u0.x - a.x, u0.y - a.y);
        z[2] = make_float2(u2.x + c.x, u2.y + c.y); z[3] = make_float2(u2.x - c.x, u2.y - c.y);
    }
    // Z[lane + 64 m] = zm[m]
    const float2 zm[4] = {z[0], z[2], z[1], z[3]};
    // Z[256 - k], k = lane + 64 m: lane (64 - lane) & 63 holds it as zm[3 - m]; lane 0 holds its own as zm[(4 - m) & 3]
    const int mir = ((64 - lane) & 63) << 2;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        float2 zc;
        zc.x = __int_as_float(__builtin_amdgcn_ds_bpermute(mir, __float_as_int(zm[3 - m].x)));
        zc.y = __int_as_float(__builtin_amdgcn_ds_bpermute(mir, __float_as_int(zm[3 - m].y)));
        if (lane == 0) zc = zm[(4 - m) & 3];
        const float2 zk = zm[m];
        const float2 E = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
        const float2 O = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y + zc.y));
        const float2 w = wu[m];
        const float2 P = make_float2(__builtin_fmaf(w.x, O.x, -(w.y * O.y)), __builtin_fmaf(w.x, O.y, w.y * O.x));
        float2 X = make_float2(E.x + P.y, E.y - P.x);
        if (m == 0 && lane == 0) X = make_float2(zk.x + zk.y, 0.f);
        const float mag = sqrtf(X.x * X.x + X.y * X.y);
        pw[wave][lane + 64 * m] = mag * mag;
    }
    if (lane == 0) {
        const float2 X = make_float2(zm[0].x - zm[0].y, 0.f);
        const float mag = sqrtf(X.x * X.x + X.y * X.y);
        pw[wave][256] = mag * mag;
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

}  // namespace lmv
