// logmel_variants.h -- copies of k_logmel (csrc/qv_layers.hip) that differ in HOW the lanes of a wave exchange FFT data, for
// tools/interference_probe.hip (round 5: which part of the log-mel kernel is it that a co-running kernel can disturb?).
// Every variant performs the SAME butterflies on the same operands in the same order as the shipped kernel, so all of
// them must reproduce its output bit for bit when nothing else runs (the probe checks that first).
//
//   k_logmel_reg    (csrc/qv_logmel_reg.h, the product's variant 1) the 256-point FFT never touches LDS memory: four complex points per lane in registers, the two
//                   in-lane strides as local butterflies, the six cross-lane strides as DPP / v_permlane*_swap exchanges;
//                   the k <-> 256 - k mirror of the real-FFT unpack through ds_bpermute (LDS crossbar, no LDS memory);
//                   power spectrum -> LDS -> mel projection as shipped
//   k_logmel_sync   the shipped kernel with one frame per 64-thread block and __syncthreads() (s_barrier + waitcnt)
//                   where the shipped one has compiler-only wave barriers
//   k_logmel_split  the shipped kernel with real / imaginary parts in separate float arrays (no 64-bit LDS accesses)
//   k_logmel_dump   the shipped kernel, additionally writing the FFT output Z[256] and the power spectrum to global
//                   memory (stage attribution: which stage's values differ first)
#pragma once

#include "../offline-tarteel_amd/csrc/qv_logmel_reg.h"   // k_logmel_reg, cmul_tw, lane_xor

namespace lmv {

// The shipped algorithm with the exchange mechanism as a parameter.
//   SYNC:  launch with 64 threads (one frame per block); every exchange point is a __syncthreads()
//   SPLIT: real and imaginary parts in separate arrays
//   DUMP:  Z and the power spectrum also go to zdump [frame][256] / pdump [frame][264]
//   UNP:   how the unpack step takes the magnitude: 0 sqrtf (shipped), 1 no square root at all (|X|^2 directly), 2 the raw
//          v_sqrt_f32 followed by 32 idle cycles before its result is used, 3 the raw v_sqrt_f32 alone, 4 sqrtf with the
//          unpack arithmetic run for every lane (no divergent branch around it), 5 sqrtf, the shipped branch, 16 idle cycles
//          in front of the EXEC write that closes it
//   DUMP additionally writes X (the real transform's bins, before the magnitude) to xdump [frame][264] and the power value
//   AS COMPUTED IN REGISTERS to rdump [frame][264] (pdump is what the next stage reads back from LDS)
template <bool SYNC, bool SPLIT, bool DUMP, int UNP = 0>
__global__ __launch_bounds__(256) void k_logmel_lds(const float *__restrict__ audio, int64_t n_max,
                                                    const int32_t *__restrict__ n_samples, const FrontendTab ft,
                                                    float *__restrict__ feats, int tm_max, float2 *__restrict__ zdump,
                                                    float *__restrict__ pdump, float2 *__restrict__ xdump = nullptr,
                                                    float *__restrict__ rdump = nullptr) {
    constexpr int NW = SYNC ? 1 : 4;
    __shared__ float2 buf[SPLIT ? 1 : NW][2][SPLIT ? 1 : 256];
    __shared__ float bre[SPLIT ? NW : 1][2][SPLIT ? 256 : 1], bim[SPLIT ? NW : 1][2][SPLIT ? 256 : 1];
    __shared__ float pw[NW][264];
    __shared__ float2 tw[256];
    const int wave = SYNC ? 0 : threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.y, t = SYNC ? blockIdx.x : blockIdx.x * 4 + wave;
    const int n = n_samples[b];
    const int tm = n / 160 + 1;
    for (int i = threadIdx.x; i < 256; i += SYNC ? 64 : 256) tw[i] = ft.twiddle[i];
    __syncthreads();
    if (t >= tm) return;
    auto fence = [&]() { if (SYNC) __syncthreads(); else __builtin_amdgcn_wave_barrier(); };
    const float *x = audio + (size_t)b * n_max;
    auto sample = [&](int i) {
        int s = t * 160 - 256 + i;
        if (s < 0) s = -s;
        if (s >= n) s = 2 * (n - 1) - s;
        s = s < 0 ? 0 : s;
        float y = x[s] - (s > 0 ? 0.97f * x[s - 1] : 0.f);
        return y * ft.window[i];
    };
    auto ld = [&](int which, int i) { return SPLIT ? make_float2(bre[wave][which][i], bim[wave][which][i]) : buf[wave][which][i]; };
    auto st = [&](int which, int i, float2 v) {
        if (SPLIT) { bre[wave][which][i] = v.x; bim[wave][which][i] = v.y; } else buf[wave][which][i] = v;
    };
    for (int i = lane; i < 256; i += 64) st(0, i, make_float2(sample(2 * i), sample(2 * i + 1)));
    int src = 0;
    for (int p = 0; p < 8; ++p) {
        int len = 1 << p;
        fence();
        for (int i = lane; i < 128; i += 64) {
            int k = i & (len - 1), j = i >> p;
            float2 u = ld(src, j * len + k), v = ld(src, j * len + k + 128);
            float2 w = tw[k * (256 >> p)];
            float2 vw = cmul_tw(v, w);
            st(src ^ 1, j * 2 * len + k, make_float2(u.x + vw.x, u.y + vw.y));
            st(src ^ 1, j * 2 * len + k + len, make_float2(u.x - vw.x, u.y - vw.y));
        }
        src ^= 1;
    }
    fence();
    const size_t frame = (size_t)b * tm_max + t;
    if (DUMP) for (int k = lane; k < 256; k += 64) zdump[frame * 256 + k] = ld(src, k);
    for (int k = lane; k < 257; k += 64) {
        float2 X;
        if (UNP == 4) {
            // the general formula for EVERY lane (bins 0 and 256 read Z[0] twice and are overwritten afterwards): the packed
            // arithmetic runs under a full EXEC mask in the four full iterations
            const int kk = k & 255;
            float2 zk = ld(src, kk), zc = ld(src, (256 - kk) & 255);
            float2 E = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
            float2 O = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y + zc.y));
            float2 w = tw[kk];
            float2 P = make_float2(__builtin_fmaf(w.x, O.x, -(w.y * O.y)), __builtin_fmaf(w.x, O.y, w.y * O.x));
            X = make_float2(E.x + P.y, E.y - P.x);
            const float2 z0 = ld(src, 0);
            if (k == 0) X = make_float2(z0.x + z0.y, 0.f);
            if (k == 256) X = make_float2(z0.x - z0.y, 0.f);
        } else if (UNP == 6) {
            // the LDS reads in front of the divergent region (pinned there), only the packed arithmetic inside it
            const int kk = k & 255;
            float2 zk = ld(src, kk), zc = ld(src, (256 - kk) & 255), w = tw[kk], z0 = ld(src, 0);
            asm volatile("" : "+v"(zk.x), "+v"(zk.y), "+v"(zc.x), "+v"(zc.y), "+v"(w.x), "+v"(w.y), "+v"(z0.x), "+v"(z0.y));
            if (k == 0 || k == 256) X = make_float2(k == 0 ? z0.x + z0.y : z0.x - z0.y, 0.f);
            else {
                float2 E = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
                float2 O = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y + zc.y));
                float2 P = make_float2(__builtin_fmaf(w.x, O.x, -(w.y * O.y)), __builtin_fmaf(w.x, O.y, w.y * O.x));
                X = make_float2(E.x + P.y, E.y - P.x);
            }
        } else
        if (k == 0 || k == 256) {
            float2 z0 = ld(src, 0);
            X = make_float2(k == 0 ? z0.x + z0.y : z0.x - z0.y, 0.f);
        } else {
            float2 zk = ld(src, k), zc = ld(src, 256 - k);
            float2 E = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
            float2 O = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y + zc.y));
            float2 w = tw[k];
            float2 P = make_float2(__builtin_fmaf(w.x, O.x, -(w.y * O.y)), __builtin_fmaf(w.x, O.y, w.y * O.x));
            X = make_float2(E.x + P.y, E.y - P.x);
            // UNP 5: the shipped branch structure, but 16 idle cycles between the branch's last (packed) operation and the
            // scalar instruction that rewrites EXEC at the end of the branch
            if (UNP == 5) {
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                f32x2_t xp = {X.x, X.y};
                asm volatile("s_nop 7\n\ts_nop 7" : "+v"(xp));
                X = make_float2(xp[0], xp[1]);
            }
        }
        const float m2 = X.x * X.x + X.y * X.y;
        float pv;
        if (UNP == 1) pv = m2;
        else if (UNP == 2) {
            float mag;
            asm volatile("v_sqrt_f32 %0, %1\n\ts_nop 15\n\ts_nop 15" : "=v"(mag) : "v"(m2));
            pv = mag * mag;
        } else if (UNP == 3) {
            float mag;
            asm volatile("v_sqrt_f32 %0, %1\n\ts_nop 0" : "=v"(mag) : "v"(m2));
            pv = mag * mag;
        } else {
            float mag = sqrtf(m2);
            pv = mag * mag;
        }
        pw[wave][k] = pv;
        if (DUMP && xdump) { xdump[frame * 264 + k] = X; rdump[frame * 264 + k] = pv; }
    }
    fence();
    if (DUMP) for (int k = lane; k < 264; k += 64) pdump[frame * 264 + k] = k < 257 ? pw[wave][k] : 0.f;
    float *out = feats + frame * QV_NMEL;
    for (int m = lane; m < QV_NMEL; m += 64) {
        int lo = ft.mel_lo[m], cnt = ft.mel_cnt[m];
        const float *w = ft.mel_w + m;
        float acc = 0.f;
        for (int k = 0; k < cnt; ++k) acc += w[k * QV_NMEL] * pw[wave][lo + k];
        out[m] = logf(acc + 5.9604644775390625e-08f);
    }
}

// per frame: how many Z values / power bins / features differ from the undisturbed run.  One wave per frame.
//   cnt[0] frames whose Z differs; cnt[1] frames whose power spectrum differs although Z does not; cnt[2] frames whose
//   features differ although the power spectrum does not; cnt[3] frames with any difference;
//   cnt[4 + i] histogram of the number of differing Z values (complex) per frame of cnt[0]: 1, 2, 3-4, 5-8, ..., 129-256
//   cnt[16 + i] the same for differing power bins per frame with any pw difference: 1, 2, 3-4, ...
//   ex[]: up to 16 examples (b, t, nz, npw, nf, first differing pw bin)
__global__ void k_classify(const float2 *__restrict__ za, const float2 *__restrict__ zb, const float *__restrict__ pa,
                           const float *__restrict__ pb, const float *__restrict__ fa, const float *__restrict__ fb,
                           int tm_max, size_t frames, unsigned long long *cnt, int *ex) {
    const size_t f = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (f >= frames) return;
    int nz = 0, np = 0, nf = 0, firstp = 1 << 20;
    for (int k = lane; k < 256; k += 64) {
        const float2 a = za[f * 256 + k], c = zb[f * 256 + k];
        nz += __float_as_uint(a.x) != __float_as_uint(c.x) || __float_as_uint(a.y) != __float_as_uint(c.y);
    }
    for (int k = lane; k < 257; k += 64)
        if (__float_as_uint(pa[f * 264 + k]) != __float_as_uint(pb[f * 264 + k])) { ++np; firstp = min(firstp, k); }
    for (int k = lane; k < QV_NMEL; k += 64) nf += __float_as_uint(fa[f * QV_NMEL + k]) != __float_as_uint(fb[f * QV_NMEL + k]);
    for (int o = 32; o; o >>= 1) {
        nz += __shfl_xor(nz, o); np += __shfl_xor(np, o); nf += __shfl_xor(nf, o); firstp = min(firstp, __shfl_xor(firstp, o));
    }
    if (lane || !(nz | np | nf)) return;
    auto bucket = [](int v) { int bkt = 0; while ((1 << bkt) < v) ++bkt; return bkt; };   // 1 -> 0, 2 -> 1, 3-4 -> 2, ...
    atomicAdd(&cnt[3], 1ull);
    if (nz) { atomicAdd(&cnt[0], 1ull); atomicAdd(&cnt[4 + bucket(nz)], 1ull); }
    else if (np) atomicAdd(&cnt[1], 1ull);
    else atomicAdd(&cnt[2], 1ull);
    if (np) atomicAdd(&cnt[16 + bucket(np)], 1ull);
    const unsigned long long slot = atomicAdd(&cnt[31], 1ull);
    if (slot < 16) {
        int *e = ex + slot * 6;
        e[0] = (int)(f / tm_max); e[1] = (int)(f % tm_max); e[2] = nz; e[3] = np; e[4] = nf; e[5] = firstp;
    }
}

// first differing stage per frame over Z -> X -> power in registers -> power read back from LDS -> features, and which
// quarter of the wave (lane >> 4 of the loop iteration k = lane + 64 m) the differing values of that stage sit in.
//   cnt[s] frames whose first differing stage is s (0 Z, 1 X, 2 register power, 3 LDS power, 4 features); cnt[8 + 4 s + q]
//   differing values of that stage in lane quarter q; cnt[31] examples taken
__global__ void k_classify5(const float2 *za, const float2 *zb, const float2 *xa, const float2 *xb, const float *ra, const float *rb,
                            const float *pa, const float *pb, const float *fa, const float *fb, size_t frames, unsigned long long *cnt) {
    const size_t f = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (f >= frames) return;
    int n[5] = {0, 0, 0, 0, 0};
    for (int k = lane; k < 256; k += 64) {
        const float2 a = za[f * 256 + k], c = zb[f * 256 + k];
        n[0] += __float_as_uint(a.x) != __float_as_uint(c.x) || __float_as_uint(a.y) != __float_as_uint(c.y);
    }
    for (int k = lane; k < 257; k += 64) {
        const float2 a = xa[f * 264 + k], c = xb[f * 264 + k];
        n[1] += __float_as_uint(a.x) != __float_as_uint(c.x) || __float_as_uint(a.y) != __float_as_uint(c.y);
        n[2] += __float_as_uint(ra[f * 264 + k]) != __float_as_uint(rb[f * 264 + k]);
        n[3] += __float_as_uint(pa[f * 264 + k]) != __float_as_uint(pb[f * 264 + k]);
    }
    for (int k = lane; k < QV_NMEL; k += 64) n[4] += __float_as_uint(fa[f * QV_NMEL + k]) != __float_as_uint(fb[f * QV_NMEL + k]);
    int tot[5];
    for (int s = 0; s < 5; ++s) { int v = n[s]; for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o); tot[s] = v; }
    int first = -1;
    for (int s = 4; s >= 0; --s) if (tot[s]) first = s;
    if (first < 0) return;
    if (lane == 0) atomicAdd(&cnt[first], 1ull);
    if (n[first]) atomicAdd(&cnt[8 + 4 * first + (lane >> 4)], (unsigned long long)n[first]);
}

}  // namespace lmv
