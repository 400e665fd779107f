// ffn_fused_bench.hip -- correctness + timing of the fused feed-forward kernel (csrc/qv_ffn.hip) against the two-kernel
// path it would replace (FFN-up k_gemm256<f16_swish> -> f16 hidden in HBM -> FFN-down residual GEMM), same inputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ffn_fused_bench.hip -o tools/ffn_fused_bench
//   tools/ffn_fused_bench [iters] [M] [in_flight for the two-kernel tile policy: 1 or 4]
// Checks: (a) 96 sampled rows against a float64 CPU evaluation of the module (hidden rounded to f16 like both device
// paths do), (b) every output of the fused kernel against the two-kernel path (they differ in summation order only).
#include "../offline-tarteel_amd/csrc/qv_gemm.hip"
#include "../offline-tarteel_amd/csrc/qv_gemm256.hip"
#define QV_FFN_ABLATIONS
#include "ffn_fused.hip"

#include <math.h>
#include <stdlib.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float frand(uint64_t &s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return ((float)((s >> 33) & 0xFFFFFF) / 8388608.0f) - 1.0f;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 50;
    const int M = argc > 2 ? atoi(argv[2]) : 8064;
    const int in_flight = argc > 3 ? atoi(argv[3]) : 1;
    uint64_t seed = 7;
    std::vector<float> w1((size_t)QV_FF * QV_D), w2((size_t)QV_D * QV_FF), b1(QV_FF), b2(QV_D), x0((size_t)M * QV_D);
    std::vector<half_t> hx((size_t)M * QV_D);
    for (auto &v : w1) v = (float)(half_t)(frand(seed) * 0.0442f);     // ~ 1/sqrt(512), f16-exact so that both paths see the same weights
    for (auto &v : w2) v = (float)(half_t)(frand(seed) * 0.0221f);     // ~ 1/sqrt(2048)
    for (auto &v : b1) v = frand(seed) * 0.1f;
    for (auto &v : b2) v = frand(seed) * 0.1f;
    for (auto &v : hx) v = (half_t)(frand(seed) * 1.7f);               // LayerNorm output: unit variance
    for (auto &v : x0) v = frand(seed) * 3.0f;
    std::vector<half_t> hw1(w1.size()), hw2(w2.size()), stream((size_t)QV_FFN_UNITS * QV_FFN_UNIT_BYTES / 2);
    for (size_t i = 0; i < w1.size(); ++i) hw1[i] = (half_t)w1[i];
    for (size_t i = 0; i < w2.size(); ++i) hw2[i] = (half_t)w2[i];
    qv_ffn_pack(w1.data(), w2.data(), stream.data());

    half_t *dX, *dW1, *dW2, *dH, *dS;
    float *dB1, *dB2, *dOut, *dOut2;
    CK(hipMalloc(&dX, hx.size() * 2)); CK(hipMalloc(&dW1, hw1.size() * 2)); CK(hipMalloc(&dW2, hw2.size() * 2));
    CK(hipMalloc(&dH, (size_t)M * QV_FF * 2)); CK(hipMalloc(&dS, stream.size() * 2));
    CK(hipMalloc(&dB1, QV_FF * 4)); CK(hipMalloc(&dB2, QV_D * 4));
    CK(hipMalloc(&dOut, x0.size() * 4)); CK(hipMalloc(&dOut2, x0.size() * 4));
    CK(hipMemcpy(dX, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW1, hw1.data(), hw1.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW2, hw2.data(), hw2.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dS, stream.data(), stream.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB1, b1.data(), QV_FF * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB2, b2.data(), QV_D * 4, hipMemcpyHostToDevice));

    FfnArgs fa = {};
    fa.X = dX; fa.ldx = QV_D; fa.Wp = (const uint8_t *)dS; fa.b1 = dB1; fa.b2 = dB2; fa.out = dOut; fa.ldo = QV_D; fa.M = M; fa.alpha = 0.5f;
    GemmArgs up = {}, dn = {};
    up.A = dX; up.W = dW1; up.bias = dB1; up.out = dH; up.M = M; up.N = QV_FF; up.K = QV_D; up.lda = QV_D; up.ldw = QV_D; up.ldo = QV_FF; up.alpha = 1.f;
    up.in_flight = in_flight;
    dn.A = dH; dn.W = dW2; dn.bias = dB2; dn.out = dOut2; dn.M = M; dn.N = QV_D; dn.K = QV_FF; dn.lda = QV_FF; dn.ldw = QV_FF; dn.ldo = QV_D; dn.alpha = 0.5f;
    dn.in_flight = in_flight;

    // ---- correctness ------------------------------------------------------------------------------------------------
    CK(hipMemcpy(dOut, x0.data(), x0.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dOut2, x0.data(), x0.size() * 4, hipMemcpyHostToDevice));
    launch_ffn_fused(fa, 0);
    launch_gemm(EPI_F16_SWISH, up, 0);
    launch_gemm(EPI_RESID, dn, 0);
    CK(hipDeviceSynchronize());
    std::vector<float> o1(x0.size()), o2(x0.size());
    CK(hipMemcpy(o1.data(), dOut, o1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o2.data(), dOut2, o2.size() * 4, hipMemcpyDeviceToHost));
    double dmax = 0, dsum = 0;
    for (size_t i = 0; i < o1.size(); ++i) { double d = fabs((double)o1[i] - o2[i]); dmax = fmax(dmax, d); dsum += d * d; }
    printf("fused vs two-kernel path (M = %d, all %zu outputs): max |d| %.3e  rms %.3e  (values ~ +-3)\n", M, o1.size(), dmax, sqrt(dsum / o1.size()));
    double cmax = 0, cmax2 = 0;
    std::vector<double> h(QV_FF);
    for (int s = 0; s < 96; ++s) {
        const int r = s < 32 ? s : s < 64 ? M - 1 - (s - 32) : (int)(((uint64_t)s * 2654435761u) % (uint64_t)M);
        for (int j = 0; j < QV_FF; ++j) {
            double a = b1[j];
            for (int k = 0; k < QV_D; ++k) a += (double)(float)hx[(size_t)r * QV_D + k] * w1[(size_t)j * QV_D + k];
            a = a / (1.0 + exp(-a));
            h[j] = (double)(float)(half_t)(float)a;
        }
        for (int n = 0; n < QV_D; ++n) {
            double a = b2[n];
            for (int j = 0; j < QV_FF; ++j) a += h[j] * w2[(size_t)n * QV_FF + j];
            const double want = x0[(size_t)r * QV_D + n] + 0.5 * a;
            cmax = fmax(cmax, fabs(want - o1[(size_t)r * QV_D + n]));
            cmax2 = fmax(cmax2, fabs(want - o2[(size_t)r * QV_D + n]));
        }
    }
    printf("against the float64 evaluation (96 rows): fused max |d| %.3e, two-kernel max |d| %.3e\n", cmax, cmax2);
    const bool ok = dmax < 5e-3 && cmax < 5e-3;
    printf("%s\n", ok ? "CORRECT" : "MISMATCH");

    // ---- timing ---------------------------------------------------------------------------------------------------------
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 2.0 * 2.0 * M * (double)QV_FF * QV_D;
    auto time = [&](const char *name, auto fn) {
        for (int i = 0; i < 5; ++i) fn();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) fn();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        printf("%-44s %8.2f us   %7.1f TFLOP/s   (M = %d)\n", name, us, flops / (us * 1e-6) / 1e12, M);
        return us;
    };
    const double t_f = time("k_ffn_fused", [&] { launch_ffn_fused(fa, 0); });
    const double t_up = time(qv_gemm_kernel_name(EPI_F16_SWISH, up), [&] { launch_gemm(EPI_F16_SWISH, up, 0); }) ;
    const double t_dn = time(qv_gemm_kernel_name(EPI_RESID, dn), [&] { launch_gemm(EPI_RESID, dn, 0); });
    const double t_2 = time("FFN-up + FFN-down back to back", [&] { launch_gemm(EPI_F16_SWISH, up, 0); launch_gemm(EPI_RESID, dn, 0); });
    // ablations (wrong results, timing only): where does a step's time go?
    time("  abl 1  no activation VALU", [&] { launch_ffn_fused_abl<1>(fa, 0); });
    time("  abl 4  no s_barrier", [&] { launch_ffn_fused_abl<4>(fa, 0); });
    time("  abl 8  no fragment reads", [&] { launch_ffn_fused_abl<8>(fa, 0); });
    time("  abl 16 no weight staging", [&] { launch_ffn_fused_abl<16>(fa, 0); });
    time("  abl 24 no reads, no staging", [&] { launch_ffn_fused_abl<24>(fa, 0); });
    time("  abl 28 no reads, staging, barrier", [&] { launch_ffn_fused_abl<28>(fa, 0); });
    time("  abl 29 MFMA + accumulate chain only", [&] { launch_ffn_fused_abl<29>(fa, 0); });
    time("  abl 93 only GEMM2's MFMAs", [&] { launch_ffn_fused_abl<93>(fa, 0); });
    time("  abl 157 only GEMM1's MFMAs", [&] { launch_ffn_fused_abl<157>(fa, 0); });
    time("  abl 221 no MFMA at all (loop skeleton)", [&] { launch_ffn_fused_abl<221>(fa, 0); });
    time("  abl 192 everything but the MFMAs", [&] { launch_ffn_fused_abl<192>(fa, 0); });
    const int blocks = (M + 127) / 128;
    printf("fused: %d blocks of 128 tokens (%d CUs busy); CU-microseconds per module: fused %.0f, two-kernel %.0f (up: 256 x 256 tiles %d, down %d)\n",
           blocks, blocks < 256 ? blocks : 256, t_f * (blocks < 256 ? blocks : 256), t_2 * 256.0, ((M + 255) / 256) * 8, ((M + 255) / 256) * 2);
    (void)t_up; (void)t_dn;
    return ok ? 0 : 1;
}
