// interference_probe.hip -- does a kernel running on one stream change what the log-mel kernel computes on another?
// Round 4 withdrew a version of k_sub01_ort (tools/withdrawn/qv_ort_conv0_mfma.hip) because, with it running in a second
// engine, k_logmel occasionally produced a few wrong bins.  This is that situation without the
// engine: stream V recomputes the log-mel features of one ragged batch over and over and counts the values that differ
// from its first (undisturbed) result; stream A runs the candidate "aggressor" (both passes of k_sub01_ort on its own
// buffers) at the same time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I offline-tarteel_amd/csrc -I include \
//         [-DPROBE_WITHDRAWN] tools/interference_probe.hip -o tools/interference_probe_{current,withdrawn}
//   tools/interference_probe_withdrawn [iterations] [aggressor: 0 none, 1 range pass, 2 real pass, 3 both, 4 the f16 path's k_sub01,
//                                                                   8 k_attention_short (64 x 126 frames), 16 k_attention_ws (64 x 376 frames)]
//                                  [victim: 0 the shipped k_logmel (+ statistics), 1-3 simple kernels, 4-6 copies of k_logmel (see k_logmel_var),
//                                   7 register FFT (logmel_variants.h), 10 one frame per block + __syncthreads(), 11 split re / im arrays,
//                                   12 shipped kernel dumping Z and the power spectrum: per-frame attribution of the differences,
//                                   13 the same with X and the register power value dumped too (five stages, lane quarters),
//                                   14 / 15 / 16 the unpack without a square root / raw v_sqrt_f32 + 32 idle cycles / raw v_sqrt_f32,
//                                   17 the unpack arithmetic for every lane (full EXEC mask in the full iterations), 18 the shipped branch with
//                                   16 idle cycles in front of the EXEC write that closes it, 19 the LDS reads in front of the branch and only the arithmetic inside,
//                                   20 / 21 FFN-up GEMM [8064,512]x[512,2048]+Swish on 256 x 256 tiles / on 128-wide tiles, 22 / 23 the
//                                   long-K GEMM [8064,2048]x[2048,512] (f32 out) on 256 x 256 / 128-wide tiles (outputs compared word by word),
//                                   99 print what the DPP / permlane primitives do on this chip and exit]
// Round 5: every run prints the device's identity (uuid, PCI bus, clocks as HIP reports them), checks that a variant
// reproduces the shipped kernel bit for bit before anything else runs, and times the victim kernel alone.
#include "../offline-tarteel_amd/csrc/qv_layers.hip"
#ifdef PROBE_WITHDRAWN
#include "withdrawn/qv_ort_conv0_mfma.hip"
#else
#include "../offline-tarteel_amd/csrc/qv_ort.hip"
#endif

#include "logmel_variants.h"
// victims 20-23: the GEMM kernels (their two translation units keep packed-FP32 instructions in the product build; only the
// `_pk` flavour of the probe compiles them as shipped)
#define sigmoidf_ sigmoidf_gemm_tu_
#include "../offline-tarteel_amd/csrc/qv_gemm.hip"
#include "../offline-tarteel_amd/csrc/qv_gemm256.hip"
#undef sigmoidf_

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_count_diff(const float *__restrict__ a, const float *__restrict__ b, size_t n, unsigned long long *cnt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long d = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) d += __float_as_uint(a[i]) != __float_as_uint(b[i]);
    if (d) atomicAdd(cnt, d);
}

// k_logmel itself with parts changed (victim 4: no wave leaves early -- frames past the end recompute the last frame;
// victim 5: twiddle factors from global memory, so no LDS table and no block barrier; victim 6: the unchanged copy)
template <int VAR>
__global__ __launch_bounds__(256) void k_logmel_var(const float *__restrict__ audio, int64_t n_max,
                                                const int32_t *__restrict__ n_samples, const FrontendTab ft,
                                                float *__restrict__ feats, int tm_max) {
    __shared__ float2 buf[4][2][256];
    __shared__ float pw[4][264];
    __shared__ float2 tw[256];   // the twiddle table: every FFT pass fetched its factors from global memory (waves parked 79 %)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.y, t = blockIdx.x * 4 + wave;
    const int n = n_samples[b];
    const int tm = n / 160 + 1;
    if (VAR != 2) { tw[threadIdx.x] = ft.twiddle[threadIdx.x]; __syncthreads(); }
    const bool past = t >= tm;
    if (past && VAR != 1) return;
    const int t_store = t;
    const int tt = past ? tm - 1 : t;
#define t tt
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
            float2 w = VAR == 2 ? ft.twiddle[k * (256 >> p)] : tw[k * (256 >> p)];
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
            float2 w = VAR == 2 ? ft.twiddle[k] : tw[k];
            float2 P = make_float2(__builtin_fmaf(w.x, O.x, -(w.y * O.y)), __builtin_fmaf(w.x, O.y, w.y * O.x));
            X = make_float2(E.x + P.y, E.y - P.x);
        }
        float mag = sqrtf(X.x * X.x + X.y * X.y);
        pw[wave][k] = mag * mag;
    }
    __builtin_amdgcn_wave_barrier();
#undef t
    if (past) return;
    float *out = feats + ((size_t)b * tm_max + t_store) * QV_NMEL;
    for (int m = lane; m < QV_NMEL; m += 64) {
        int lo = ft.mel_lo[m], cnt = ft.mel_cnt[m];
        const float *w = ft.mel_w + m;        // tap-major [32][80]
        float acc = 0.f;
        for (int k = 0; k < cnt; ++k) acc += w[k * QV_NMEL] * pw[wave][lo + k];
        out[m] = logf(acc + 5.9604644775390625e-08f);
    }
}


// other victims: which kind of kernel can be disturbed?  1: registers only (a dependent FMA / sqrt / log chain per thread);
// 2: lanes exchange values through LDS without a barrier (wave-synchronous, like k_logmel's FFT); 3: the same exchange
// with __syncthreads() between the write and the read
template <int KIND>
__global__ __launch_bounds__(256) void k_victim(const float *__restrict__ x, float *__restrict__ y, size_t n) {
    __shared__ float sh[4][64];
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (size_t i = i0; i < n; i += (size_t)gridDim.x * 256) {
        float v = x[i], a = 0.5f;
        for (int r = 0; r < 24; ++r) {
            a = __builtin_fmaf(a, 0.75f, v);
            a = sqrtf(a * a + 1.0f);
            if (KIND >= 2) {
                __builtin_amdgcn_wave_barrier();
                sh[wave][lane] = a;
                if (KIND == 3) __syncthreads(); else __builtin_amdgcn_wave_barrier();
                a += 0.25f * sh[wave][(lane + 1 + r) & 63];
                if (KIND == 3) __syncthreads();
            }
        }
        y[i] = logf(a);
    }
}

// what the cross-lane primitives do (ground truth for the register FFT): out[i][lane]
__global__ void k_lane_primitives(int *out) {
    const int lane = threadIdx.x;
    out[0 * 64 + lane] = __builtin_amdgcn_mov_dpp(lane, 0x104, 0xf, 0xf, true);   // row_shl:4
    out[1 * 64 + lane] = __builtin_amdgcn_mov_dpp(lane, 0x114, 0xf, 0xf, true);   // row_shr:4
    out[2 * 64 + lane] = __builtin_amdgcn_mov_dpp(lane, 0x13C, 0xf, 0xf, true);   // wave_ror:1
    out[3 * 64 + lane] = __builtin_amdgcn_mov_dpp(lane, 0x134, 0xf, 0xf, true);   // wave_rol:1
    out[4 * 64 + lane] = __builtin_amdgcn_mov_dpp(lane, 0x140, 0xf, 0xf, true);   // row_mirror
    out[5 * 64 + lane] = __builtin_amdgcn_mov_dpp(lane, 0x141, 0xf, 0xf, true);   // row_half_mirror
    const auto a = __builtin_amdgcn_permlane16_swap((unsigned)lane, (unsigned)(lane + 100), false, false);
    out[6 * 64 + lane] = (int)a[0]; out[7 * 64 + lane] = (int)a[1];
    const auto c = __builtin_amdgcn_permlane32_swap((unsigned)lane, (unsigned)(lane + 100), false, false);
    out[8 * 64 + lane] = (int)c[0]; out[9 * 64 + lane] = (int)c[1];
    out[10 * 64 + lane] = __builtin_amdgcn_mov_dpp(lane, 0x128, 0xf, 0xf, true);  // row_ror:8
}

__global__ void k_copy_f4(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

template <class T> static T *dev(const std::vector<T> &h) {
    T *p = nullptr;
    if (hipMalloc(&p, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    (void)hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return p;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400, aggr = argc > 2 ? atoi(argv[2]) : 3, victim = argc > 3 ? atoi(argv[3]) : 0,
              wfrac = argc > 4 ? atoi(argv[4]) : 0;   // 1: the aggressor's conv.0 weights are fractions in (-1, 1) instead of integers
    {
        hipDeviceProp_t pr;
        CK(hipGetDeviceProperties(&pr, 0));
        printf("device %s uuid ", pr.gcnArchName);
        for (int i = 0; i < 16; ++i) printf("%02x", (unsigned char)pr.uuid.bytes[i]);
        printf(" pci %04x:%02x:%02x clock %d kHz mem clock %d kHz CUs %d\n", pr.pciDomainID, pr.pciBusID, pr.pciDeviceID, pr.clockRate,
               pr.memoryClockRate, pr.multiProcessorCount);
    }
    if (victim == 99) {
        int *d_o;
        CK(hipMalloc(&d_o, 11 * 64 * 4));
        hipLaunchKernelGGL(k_lane_primitives, dim3(1), dim3(64), 0, 0, d_o);
        std::vector<int> h(11 * 64);
        CK(hipMemcpy(h.data(), d_o, h.size() * 4, hipMemcpyDeviceToHost));
        const char *nm[11] = {"row_shl:4", "row_shr:4", "wave_ror:1", "wave_rol:1", "row_mirror", "row_half_mirror", "permlane16_swap(lane, lane + 100) [0]",
                              "permlane16_swap [1]", "permlane32_swap(lane, lane + 100) [0]", "permlane32_swap [1]", "row_ror:8"};
        for (int i = 0; i < 11; ++i) {
            printf("%-40s", nm[i]);
            for (int l = 0; l < 64; ++l) printf(" %d", h[i * 64 + l]);
            printf("\n");
        }
        return 0;
    }
    const int B = 58;
    const int64_t n_max = 480000;
    srand(7);
    std::vector<int32_t> n(B), tm(B), l1(B), l2(B);
    int tm_max = 0, t2_max = 0;
    for (int b = 0; b < B; ++b) {
        n[b] = b == 0 ? (int)n_max - 123 : 800 + rand() % (int)(n_max - 800);
        tm[b] = n[b] / 160 + 1; l1[b] = (tm[b] - 1) / 2 + 1; l2[b] = (l1[b] - 1) / 2 + 1;
        tm_max = std::max(tm_max, tm[b]); t2_max = std::max(t2_max, l2[b]);
    }
    std::vector<float> audio((size_t)B * n_max);
    uint32_t st = 12345;
    for (auto &x : audio) { st = st * 1664525u + 1013904223u; x = ((int)(st >> 8) % 2001 - 1000) * 2.5e-4f; }
    for (int b = 0; b < B; ++b) for (int64_t i = n[b]; i < n_max; ++i) audio[(size_t)b * n_max + i] = 0.f;
    // front-end tables: shape matters, contents only have to be deterministic
    std::vector<float> window(512, 0.f), melw(32 * 80, 0.f);
    std::vector<float2> tw(256);
    std::vector<int32_t> lo(80), cnt(80);
    for (int i = 0; i < 400; ++i) window[56 + i] = 0.5f - 0.5f * cosf(6.2831853f * i / 399.f);
    for (int m = 0; m < 256; ++m) tw[m] = make_float2(cosf(-6.2831853f * m / 512.f), sinf(-6.2831853f * m / 512.f));
    for (int m = 0; m < 80; ++m) {
        lo[m] = 1 + m * 2; cnt[m] = 4 + m / 4;
        for (int k = 0; k < cnt[m]; ++k) melw[k * 80 + m] = 1.f - fabsf(2.f * k / (cnt[m] - 1) - 1.f) + 1e-3f;
    }
    FrontendTab ft{dev(window), dev(tw), dev(lo), dev(cnt), dev(melw)};
    float *d_audio = dev(audio);
    int32_t *d_n = dev(n), *d_tm = dev(tm), *d_l1 = dev(l1), *d_l2 = dev(l2);
    const size_t nf = (size_t)B * tm_max * QV_NMEL;
    float *feats_v, *feats_ref, *feats_a, *out_a;
    double *stats_v, *stats_a;
    uint32_t *mm;
    unsigned long long *d_cnt;
    CK(hipMalloc(&feats_v, nf * 4)); CK(hipMalloc(&feats_ref, nf * 4)); CK(hipMalloc(&feats_a, nf * 4));
    CK(hipMalloc(&out_a, (size_t)B * t2_max * 20 * QV_SUBC * 4));
    CK(hipMalloc(&stats_v, qv_melstats_doubles(B) * 8)); CK(hipMalloc(&stats_a, qv_melstats_doubles(B) * 8));
    CK(hipMalloc(&mm, (size_t)3 * B * QV_MM_STRIDE * 4)); CK(hipMalloc(&d_cnt, 8));
    CK(hipMemset(d_cnt, 0, 8));
    std::vector<float> w0(9 * QV_SUBC), w1(9 * QV_SUBC), b0(QV_SUBC), b1(QV_SUBC);
    for (auto &x : w0) x = wfrac ? (rand() % 20001 - 10000) * 1e-4f : (float)(rand() % 255 - 127);
    for (auto &x : w1) x = (float)(rand() % 255 - 127);
    for (auto &x : b0) x = (rand() % 2001 - 1000) * 1e-3f;
    for (auto &x : b1) x = (rand() % 2001 - 1000) * 1e-3f;
    float *d_w0 = dev(w0), *d_w1 = dev(w1), *d_b0 = dev(b0), *d_b1 = dev(b1);
    // attention inputs (tools/att_bench.hip's): 64 utterances of 126 / 376 frames
    const int AB = 64, AT = (aggr & 16) ? 376 : 126, at_pad = (AT + 31) / 32 * 32, apos_ld = 17 * QV_D;
    std::vector<half_t> hqk((size_t)AB * AT * 2 * QV_D), hvt((size_t)AB * QV_D * at_pad), hpos((size_t)(2 * AT - 1) * apos_ld);
    for (auto &x : hqk) x = (half_t)((rand() % 2001 - 1000) / 1000.0f);
    for (auto &x : hvt) x = (half_t)((rand() % 2001 - 1000) / 1000.0f);
    for (auto &x : hpos) x = (half_t)((rand() % 2001 - 1000) / 1000.0f);
    std::vector<float> hbias(QV_D, 0.01f);
    std::vector<int32_t> alen(AB, AT), aoff(AB);
    for (int b = 0; b < AB; ++b) aoff[b] = b * AT;
    half_t *a_qk = dev(hqk), *a_vt = dev(hvt), *a_pos = dev(hpos), *a_out = nullptr;
    float *a_bu = dev(hbias), *a_bv = dev(hbias);
    int32_t *a_len = dev(alen), *a_off = dev(aoff);
    CK(hipMalloc(&a_out, (size_t)AB * AT * QV_D * 2));
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    // undisturbed reference + the aggressor's own inputs
    const size_t nv = 58ull * 3000 * 80;     // the other victims work on the first values of the audio, same output size
    // victim 12: Z and the power spectrum of every frame, undisturbed (ref) and in the loop (v)
    float2 *z_ref = nullptr, *z_v = nullptr;
    float *p_ref = nullptr, *p_v = nullptr;
    unsigned long long *d_cls = nullptr;
    int *d_ex = nullptr;
    const size_t frames = (size_t)B * tm_max;
    float2 *x_ref = nullptr, *x_v = nullptr;
    float *r_ref = nullptr, *r_v = nullptr;
    if (victim == 13) {
        CK(hipMalloc(&x_ref, frames * 264 * 8)); CK(hipMalloc(&x_v, frames * 264 * 8));
        CK(hipMalloc(&r_ref, frames * 264 * 4)); CK(hipMalloc(&r_v, frames * 264 * 4));
        CK(hipMemset(x_ref, 0, frames * 264 * 8)); CK(hipMemset(x_v, 0, frames * 264 * 8));
        CK(hipMemset(r_ref, 0, frames * 264 * 4)); CK(hipMemset(r_v, 0, frames * 264 * 4));
    }
    if (victim == 12 || victim == 13) {
        CK(hipMalloc(&z_ref, frames * 256 * 8)); CK(hipMalloc(&z_v, frames * 256 * 8));
        CK(hipMalloc(&p_ref, frames * 264 * 4)); CK(hipMalloc(&p_v, frames * 264 * 4));
        CK(hipMalloc(&d_cls, 32 * 8)); CK(hipMalloc(&d_ex, 16 * 6 * 4));
        CK(hipMemset(z_ref, 0, frames * 256 * 8)); CK(hipMemset(z_v, 0, frames * 256 * 8));
        CK(hipMemset(p_ref, 0, frames * 264 * 4)); CK(hipMemset(p_v, 0, frames * 264 * 4));
        CK(hipMemset(d_cls, 0, 32 * 8)); CK(hipMemset(d_ex, 0, 16 * 6 * 4));
    }
    // GEMM victims: operands of gemm_bench's kind, outputs into the feature buffers (13.9 M words >= 8064 x 2048 halves)
    half_t *g_A = nullptr, *g_W = nullptr;
    float *g_bias = nullptr;
    uint8_t *g_W8 = nullptr;
    float *g_w8s = nullptr, *g_res0 = nullptr;
    if (victim >= 20 && victim <= 25) {
        const int GM = 8064;
        std::vector<half_t> hA((size_t)GM * 2048), hW((size_t)2048 * 2048);
        uint32_t gs = 99;
        for (auto &x : hA) { gs = gs * 1664525u + 1013904223u; x = (half_t)(((int)(gs >> 9) % 2001 - 1000) * 1e-3f); }
        for (auto &x : hW) { gs = gs * 1664525u + 1013904223u; x = (half_t)(((int)(gs >> 9) % 2001 - 1000) * 5e-5f); }
        std::vector<float> hb(4096);
        for (auto &x : hb) { gs = gs * 1664525u + 1013904223u; x = ((int)(gs >> 9) % 2001 - 1000) * 1e-4f; }
        g_A = dev(hA); g_W = dev(hW); g_bias = dev(hb);
        if (victim >= 24) {   // the two instantiations that KEEP packed FP32 in the product (k_gemm_pk: W8A16 on 128-wide tiles, GLU / residual epilogue)
            std::vector<uint8_t> h8((size_t)1024 * 512);
            for (auto &x : h8) { gs = gs * 1664525u + 1013904223u; x = (uint8_t)(1 + (gs >> 9) % 255); }
            std::vector<float> hs(1024), hr((size_t)GM * 512);
            for (auto &x : hs) { gs = gs * 1664525u + 1013904223u; x = (1 + (int)(gs >> 9) % 1000) * 1e-5f; }
            for (auto &x : hr) { gs = gs * 1664525u + 1013904223u; x = ((int)(gs >> 9) % 2001 - 1000) * 1e-3f; }
            g_W8 = dev(h8); g_w8s = dev(hs); g_res0 = dev(hr);
        }
    }
    auto run_gemm_victim = [&](float *out) {
        GemmArgs g = {};
        if (victim >= 24) {
            const bool glu = victim == 24;
            g.A = g_A; g.W8 = g_W8; g.w8scale = g_w8s; g.bias = g_bias; g.out = out;
            g.M = 8064; g.N = glu ? 1024 : 512; g.K = 512; g.lda = 512; g.ldw = 512; g.ldo = 512; g.alpha = glu ? 1.f : 0.5f;
            // the residual epilogue reads and rewrites its output: start every launch from the same residual stream
            if (!glu) hipLaunchKernelGGL(k_copy_f4, dim3(2048), dim3(256), 0, sv, (const float4 *)g_res0, (float4 *)out, (size_t)8064 * 512 / 4);
            qv_gemm_set_t256(0);
            launch_gemm(glu ? EPI_GLU : EPI_RESID, g, sv);
            return;
        }
        const bool up = victim <= 21;
        g.A = g_A; g.W = g_W; g.bias = g_bias; g.out = out;
        g.M = 8064; g.N = up ? 2048 : 512; g.K = up ? 512 : 2048; g.lda = g.K; g.ldw = g.K; g.ldo = g.N; g.alpha = 1.f;
        qv_gemm_set_t256((victim & 1) ? 0 : 2);
        launch_gemm(up ? EPI_F16_SWISH : EPI_F32, g, sv);
    };
    auto run_victim = [&](float *out) {
        const dim3 g4((tm_max + 3) / 4, B);
        if (victim >= 20 && victim <= 25) { run_gemm_victim(out); return; }
        if (victim == 0) launch_logmel(d_audio, n_max, d_n, ft, out, tm_max, stats_v, B, sv);
        else if (victim == 7) hipLaunchKernelGGL(lmv::k_logmel_reg<0>, g4, dim3(256), 0, sv, d_audio, n_max, d_n, ft, out, tm_max);
        else if (victim == 10) hipLaunchKernelGGL((lmv::k_logmel_lds<true, false, false>), dim3(tm_max, B), dim3(64), 0, sv, d_audio, n_max, d_n, ft, out, tm_max, (float2 *)nullptr, (float *)nullptr);
        else if (victim == 11) hipLaunchKernelGGL((lmv::k_logmel_lds<false, true, false>), g4, dim3(256), 0, sv, d_audio, n_max, d_n, ft, out, tm_max, (float2 *)nullptr, (float *)nullptr);
        else if (victim == 12) hipLaunchKernelGGL((lmv::k_logmel_lds<false, false, true>), g4, dim3(256), 0, sv, d_audio, n_max, d_n, ft, out, tm_max,
                                                  out == feats_ref ? z_ref : z_v, out == feats_ref ? p_ref : p_v);
        else if (victim == 13) hipLaunchKernelGGL((lmv::k_logmel_lds<false, false, true>), g4, dim3(256), 0, sv, d_audio, n_max, d_n, ft, out, tm_max,
                                                  out == feats_ref ? z_ref : z_v, out == feats_ref ? p_ref : p_v, out == feats_ref ? x_ref : x_v,
                                                  out == feats_ref ? r_ref : r_v);
        else if (victim == 19) hipLaunchKernelGGL((lmv::k_logmel_lds<false, false, false, 6>), g4, dim3(256), 0, sv, d_audio, n_max, d_n, ft, out, tm_max, (float2 *)nullptr, (float *)nullptr, (float2 *)nullptr, (float *)nullptr);
        else if (victim == 18) hipLaunchKernelGGL((lmv::k_logmel_lds<false, false, false, 5>), g4, dim3(256), 0, sv, d_audio, n_max, d_n, ft, out, tm_max, (float2 *)nullptr, (float *)nullptr, (float2 *)nullptr, (float *)nullptr);
        else if (victim == 17) hipLaunchKernelGGL((lmv::k_logmel_lds<false, false, false, 4>), g4, dim3(256), 0, sv, d_audio, n_max, d_n, ft, out, tm_max, (float2 *)nullptr, (float *)nullptr, (float2 *)nullptr, (float *)nullptr);
        else if (victim >= 14 && victim <= 16) {     // the unpack's square root: none / raw + 32 idle cycles / raw
            if (victim == 14) hipLaunchKernelGGL((lmv::k_logmel_lds<false, false, false, 1>), g4, dim3(256), 0, sv, d_audio, n_max, d_n, ft, out, tm_max, (float2 *)nullptr, (float *)nullptr, (float2 *)nullptr, (float *)nullptr);
            if (victim == 15) hipLaunchKernelGGL((lmv::k_logmel_lds<false, false, false, 2>), g4, dim3(256), 0, sv, d_audio, n_max, d_n, ft, out, tm_max, (float2 *)nullptr, (float *)nullptr, (float2 *)nullptr, (float *)nullptr);
            if (victim == 16) hipLaunchKernelGGL((lmv::k_logmel_lds<false, false, false, 3>), g4, dim3(256), 0, sv, d_audio, n_max, d_n, ft, out, tm_max, (float2 *)nullptr, (float *)nullptr, (float2 *)nullptr, (float *)nullptr);
        }
        else if (victim >= 4) {
            const dim3 g((tm_max + 3) / 4, B);
            if (victim == 4) hipLaunchKernelGGL(k_logmel_var<1>, g, dim3(256), 0, sv, d_audio, n_max, d_n, ft, out, tm_max);
            else if (victim == 5) hipLaunchKernelGGL(k_logmel_var<2>, g, dim3(256), 0, sv, d_audio, n_max, d_n, ft, out, tm_max);
            else hipLaunchKernelGGL(k_logmel_var<0>, g, dim3(256), 0, sv, d_audio, n_max, d_n, ft, out, tm_max);
        }
        else if (victim == 1) hipLaunchKernelGGL(k_victim<1>, dim3(4096), dim3(256), 0, sv, d_audio, out, std::min(nv, nf));
        else if (victim == 2) hipLaunchKernelGGL(k_victim<2>, dim3(4096), dim3(256), 0, sv, d_audio, out, std::min(nv, nf) / 256 * 256);
        else hipLaunchKernelGGL(k_victim<3>, dim3(4096), dim3(256), 0, sv, d_audio, out, std::min(nv, nf) / 256 * 256);
    };
    CK(hipMemset(feats_ref, 0, nf * 4)); CK(hipMemset(feats_v, 0, nf * 4));
    if (victim) run_victim(feats_ref); else
    launch_logmel(d_audio, n_max, d_n, ft, feats_ref, tm_max, stats_v, B, sv);
    launch_logmel(d_audio, n_max, d_n, ft, feats_a, tm_max, stats_a, B, sv);
    CK(hipStreamSynchronize(sv));
    if (victim >= 4 && !(victim >= 14 && victim <= 16) && victim < 20) {     // a copy of k_logmel has to reproduce the shipped kernel bit for bit (feats_a is the shipped kernel's output)
        hipLaunchKernelGGL(k_count_diff, dim3(1024), dim3(256), 0, sv, feats_ref, feats_a, nf, d_cnt);
        CK(hipStreamSynchronize(sv));
        unsigned long long c;
        CK(hipMemcpy(&c, d_cnt, 8, hipMemcpyDeviceToHost));
        printf("victim %d against the shipped k_logmel, nothing else running: %llu differing values of %zu\n", victim, c, nf);
        CK(hipMemset(d_cnt, 0, 8));
    }
    {   // the victim kernel alone: average of 20 launches
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        run_victim(feats_v);
        CK(hipEventRecord(e0, sv));
        for (int i = 0; i < 20; ++i) run_victim(feats_v);
        CK(hipEventRecord(e1, sv));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("victim %d alone: %.1f us per launch (%d clips, %d frames max%s)\n", victim, ms * 50.f, B, tm_max, victim == 0 ? ", with the statistics kernels" : "");
    }
    uint32_t *mm_mel = mm, *mm_c0 = mm + (size_t)B * QV_MM_STRIDE, *mm_c1 = mm + (size_t)2 * B * QV_MM_STRIDE;
    unsigned long long last = 0;
    int bad_iters = 0;
    for (int it = 0; it < iters; ++it) {
        if (aggr) {
            if (aggr & 3) {
                launch_mm_init(mm, (size_t)3 * B * QV_MM_STRIDE, sa);
                launch_mel_minmax(feats_a, d_n, tm_max, stats_a, mm_mel, B, sa);
            }
            if (aggr & 1) launch_sub01_ort(0, feats_a, tm_max, d_tm, stats_a, d_w0, 0.01f, d_b0, d_l1, d_w1, 0.01f, d_b1, d_l2, mm_mel, mm_c0, mm_c1, out_a, t2_max, B, sa);
            if (aggr & 24)
                for (int r = 0; r < 17; ++r) launch_attention(a_qk, a_vt, a_pos, apos_ld, a_bu, a_bv, a_len, a_off, a_out, AT, AT, at_pad, AB, sa);
            if (aggr & 4) launch_sub01(feats_a, tm_max, d_tm, stats_a, d_w0, d_b0, d_l1, d_w1, d_b1, (half_t *)out_a, t2_max, B, sa);
            if (aggr & 2) launch_sub01_ort(1, feats_a, tm_max, d_tm, stats_a, d_w0, 0.01f, d_b0, d_l1, d_w1, 0.01f, d_b1, d_l2, mm_mel, mm_c0, mm_c1, out_a, t2_max, B, sa);
        }
        run_victim(feats_v);
        hipLaunchKernelGGL(k_count_diff, dim3(1024), dim3(256), 0, sv, feats_v, feats_ref, nf, d_cnt);
        if (victim == 12)
            hipLaunchKernelGGL(lmv::k_classify, dim3((unsigned)((frames + 3) / 4)), dim3(256), 0, sv, z_v, z_ref, p_v, p_ref, feats_v, feats_ref, tm_max, frames, d_cls, d_ex);
        if (victim == 13)
            hipLaunchKernelGGL(lmv::k_classify5, dim3((unsigned)((frames + 3) / 4)), dim3(256), 0, sv, z_v, z_ref, x_v, x_ref, r_v, r_ref, p_v, p_ref, feats_v, feats_ref, frames, d_cls);
        if (it % 8 == 7 || it == iters - 1) {
            CK(hipStreamSynchronize(sv));
            unsigned long long c;
            CK(hipMemcpy(&c, d_cnt, 8, hipMemcpyDeviceToHost));
            if (c != last) { ++bad_iters; last = c; }
        }
    }
    CK(hipDeviceSynchronize());
    printf("%s k_sub01_ort, aggressor passes %d, victim %d, %d iterations of k_logmel on %d clips (%d frames max): %llu differing values, "
           "seen in %d of %d checked groups of 8 iterations\n",
#ifdef PROBE_WITHDRAWN
           "WITHDRAWN",
#else
           "current",
#endif
           aggr, victim, iters, B, tm_max, last, bad_iters, (iters + 7) / 8);
    if (victim == 13) {
        unsigned long long c[32];
        CK(hipMemcpy(c, d_cls, sizeof c, hipMemcpyDeviceToHost));
        const char *st[5] = {"FFT output Z", "X (real transform, before the magnitude)", "power value in registers", "power value read back from LDS", "features"};
        for (int i = 0; i < 5; ++i)
            printf("first differing stage %-45s %8llu frames; its differing values by lane quarter (0-15, 16-31, 32-47, 48-63): %llu %llu %llu %llu\n", st[i],
                   c[i], c[8 + 4 * i], c[9 + 4 * i], c[10 + 4 * i], c[11 + 4 * i]);
    }
    if (victim == 12) {
        unsigned long long c[32];
        int ex[96];
        CK(hipMemcpy(c, d_cls, sizeof c, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ex, d_ex, sizeof ex, hipMemcpyDeviceToHost));
        printf("attribution: %llu frames differ; %llu already in the FFT output Z, %llu first in the power spectrum, %llu first in the features\n",
               c[3], c[0], c[1], c[2]);
        printf("  differing Z values per such frame (1, 2, 3-4, 5-8, ... 129-256):");
        for (int i = 0; i < 9; ++i) printf(" %llu", c[4 + i]);
        printf("\n  differing power bins per frame (1, 2, 3-4, 5-8, ...):");
        for (int i = 0; i < 10; ++i) printf(" %llu", c[16 + i]);
        printf("\n");
        for (int i = 0; i < 16 && (unsigned long long)i < c[31]; ++i)
            printf("  e.g. clip %d frame %d: %d Z values, %d power bins (first %d), %d features\n", ex[i * 6], ex[i * 6 + 1], ex[i * 6 + 2], ex[i * 6 + 3],
                   ex[i * 6 + 5], ex[i * 6 + 4]);
    }
    return 0;
}
