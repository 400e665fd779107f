// gemm_bench.hip -- standalone timing + correctness harness for csrc/qv_gemm.hip (dev tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/gemm_bench.hip -o tools/gemm_bench
//   tools/gemm_bench [iters] [M] [t256 mode: 0 = 128-wide tiles, 1 = default policy, 2 = 256 x 256 wherever N % 256 == 0,
//                                 3 = like 2 with the four-wave kernel k_gemm256q (128 x 128 per wave)]
// Every shape the 256 x 256 kernel accepts is also compared BIT FOR BIT with the 128-wide kernel (all outputs).
#define QV_GEMM_Q_VARIANT "../../tools/gemm256q.h"
#include "../offline-tarteel_amd/csrc/qv_gemm.hip"
#include "../offline-tarteel_amd/csrc/qv_gemm256.hip"

#include <math.h>
#include <stdlib.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float frand(uint64_t &s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return ((float)((s >> 33) & 0xFFFFFF) / 8388608.0f) - 1.0f;
}

int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 50;
    const int M = argc > 2 ? atoi(argv[2]) : 8064;
    const int mode_arg = argc > 3 ? atoi(argv[3]) : 1;
    const int mode = mode_arg == 3 ? 2 : mode_arg;
    qv_gemm_set_q(mode_arg == 3 ? 1 : 0);
    struct Sh { const char *name; int epi, N, K, ldo; float alpha; } shapes[] = {
        {"ff_up    swish N2048 K512 ", EPI_F16_SWISH, 2048, 512, 2048, 1.f},
        {"ff_down  resid N512 K2048 ", EPI_RESID, 512, 2048, 512, 0.5f},
        {"qkv            N1536 K512 ", EPI_QKV, 1536, 512, 1024, 1.f},
        {"out/pw2  resid N512 K512  ", EPI_RESID, 512, 512, 512, 1.f},
        {"pw1      glu   N1024 K512 ", EPI_GLU, 1024, 512, 512, 1.f},
        {"head     f32   N1152 K512 ", EPI_F32, 1152, 512, 1152, 1.f},
        {"sub_out  f32   N512 K2560 ", EPI_F32, 512, 2560, 512, 1.f},
        {"ff_up shape, ReLU epilogue", EPI_F16_RELU, 2048, 512, 2048, 1.f},   // ablation: what the Swish costs
    };
    uint64_t seed = 1;
    size_t maxA = (size_t)M * 2560, maxW = (size_t)2048 * 2560, maxO = (size_t)M * 2048;
    std::vector<half_t> hA(maxA), hW(maxW);
    for (auto &v : hA) v = (half_t)frand(seed);
    for (auto &v : hW) v = (half_t)(frand(seed) * 0.05f);
    std::vector<float> hb(4096);
    for (auto &v : hb) v = frand(seed) * 0.1f;
    half_t *dA, *dW, *dV;
    float *db;
    void *dO;
    CK(hipMalloc(&dA, maxA * 2)); CK(hipMalloc(&dW, maxW * 2)); CK(hipMalloc(&dO, maxO * 4)); CK(hipMalloc(&db, 4096 * 4));
    CK(hipMalloc(&dV, (size_t)(M / 126 + 1) * 512 * 128 * 2));
    CK(hipMemcpy(dA, hA.data(), maxA * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), maxW * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), 4096 * 4, hipMemcpyHostToDevice));
    // packed-row owner map for the QKV epilogue: utterances of 126 frames
    std::vector<int32_t> hmap(M);
    for (int r = 0; r < M; ++r) hmap[r] = ((r / 126) << 16) | (r % 126);
    int32_t *dmap;
    CK(hipMalloc(&dmap, (size_t)M * 4));
    CK(hipMemcpy(dmap, hmap.data(), (size_t)M * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double tot_ms = 0, tot_fl = 0;
    const double per_layer[] = {2, 2, 1, 2, 1, 0, 0, 0};
    for (auto &sh : shapes) {
        GemmArgs g = {};
        g.A = dA; g.W = dW; g.bias = db; g.out = dO; g.out2 = dV;
        g.M = M; g.N = sh.N; g.K = sh.K; g.lda = sh.K; g.ldw = sh.K; g.ldo = sh.ldo; g.alpha = sh.alpha;
        g.t_max = 126; g.t_pad = 128; g.row_map = dmap;
        const size_t vbytes = (size_t)(M / 126 + 1) * 512 * 128 * 2;
        long ndiff = -1;
        if (sh.N % 256 == 0) {
            // bit-for-bit: 128-wide tiles vs 256 x 256 tiles, every output byte (plus Vt for the QKV epilogue)
            const size_t obytes = (size_t)M * sh.ldo * ((sh.epi == EPI_F32 || sh.epi == EPI_RESID) ? 4 : 2);
            std::vector<unsigned char> o0(obytes), o1(obytes), v0(vbytes), v1(vbytes);
            for (int pass = 0; pass < 2; ++pass) {
                CK(hipMemset(dO, 0, maxO * 4));
                CK(hipMemset(dV, 0, vbytes));
                qv_gemm_set_t256(pass ? 2 : 0);
                launch_gemm(sh.epi, g, 0);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(pass ? o1.data() : o0.data(), dO, obytes, hipMemcpyDeviceToHost));
                CK(hipMemcpy(pass ? v1.data() : v0.data(), dV, vbytes, hipMemcpyDeviceToHost));
            }
            ndiff = 0;
            for (size_t i = 0; i < obytes; ++i) ndiff += o0[i] != o1[i];
            if (sh.epi == EPI_QKV) for (size_t i = 0; i < vbytes; ++i) ndiff += v0[i] != v1[i];
        }
        qv_gemm_set_t256(mode);
        CK(hipMemset(dO, 0, maxO * 4));
        launch_gemm(sh.epi, g, 0);
        CK(hipDeviceSynchronize());
        // spot check 64 entries against a host dot product
        double maxerr = 0;
        if (sh.epi == EPI_F32 || sh.epi == EPI_RESID) {
            std::vector<float> ho((size_t)M * sh.ldo);
            CK(hipMemcpy(ho.data(), dO, ho.size() * 4, hipMemcpyDeviceToHost));
            for (int t = 0; t < 64; ++t) {
                int r = (t * 977) % M, c = (t * 131) % (sh.N < 1025 ? sh.N : 1025);
                double acc = 0;
                for (int k = 0; k < sh.K; ++k) acc += (double)(float)hA[(size_t)r * sh.K + k] * (double)(float)hW[(size_t)c * sh.K + k];
                double want = sh.alpha * (acc + hb[c]);
                maxerr = fmax(maxerr, fabs(want - ho[(size_t)r * sh.ldo + c]));
            }
        }
        if (sh.epi == EPI_F16_SWISH) {
            std::vector<half_t> ho((size_t)M * sh.ldo);
            CK(hipMemcpy(ho.data(), dO, ho.size() * 2, hipMemcpyDeviceToHost));
            for (int t = 0; t < 256; ++t) {
                int r = (t * 977) % M, c = (t * 131) % sh.N;
                double acc = 0;
                for (int k = 0; k < sh.K; ++k) acc += (double)(float)hA[(size_t)r * sh.K + k] * (double)(float)hW[(size_t)c * sh.K + k];
                double x = acc + hb[c], want = x / (1.0 + exp(-x));
                maxerr = fmax(maxerr, fabs(want - (double)(float)ho[(size_t)r * sh.ldo + c]));
            }
        }
        for (int i = 0; i < 3; ++i) launch_gemm(sh.epi, g, 0);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) launch_gemm(sh.epi, g, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double us = ms * 1e3 / iters, fl = 2.0 * M * sh.N * sh.K;
        printf("%s %8.2f us  %7.1f TF/s  maxerr %.2e  bytes differing 128-wide vs 256x256: %ld\n", sh.name, us, fl / us / 1e6, maxerr, ndiff);
        int idx = (int)(&sh - shapes);
        tot_ms += per_layer[idx] * us;
        tot_fl += per_layer[idx] * fl;
    }
    printf("per-layer GEMM time %.1f us -> %.1f TF/s; x17 = %.2f ms\n", tot_ms, tot_fl / tot_ms / 1e6, tot_ms * 17 / 1e3);
    return 0;
}
