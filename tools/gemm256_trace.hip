// gemm256_trace.hip -- dev tool: ablations, block timeline and per-K-tile stamps of k_gemm256 (qv_gemm256.hip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DQV_GEMM_TRACE tools/gemm256_trace.hip -o tools/gemm256_trace
//   tools/gemm256_trace [M]
#include "../offline-tarteel_amd/csrc/qv_gemm.hip"
#include "../offline-tarteel_amd/csrc/qv_gemm256.hip"

#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 8064;
    struct Sh { const char *name; int epi, N, K, ldo; float alpha; } shapes[] = {
        {"ff_up   N2048 K512 ", EPI_F16_SWISH, 2048, 512, 2048, 1.f},
        {"ff_down N512 K2048 ", EPI_RESID, 512, 2048, 512, 0.5f},
        {"pw1 glu N1024 K512 ", EPI_GLU, 1024, 512, 512, 1.f},
        {"ff_up, plain f16 epilogue ", EPI_F16, 2048, 512, 2048, 1.f},
    };
    half_t *dA, *dW; float *db; void *dO;
    CK(hipMalloc(&dA, (size_t)M * 2560 * 2)); CK(hipMalloc(&dW, (size_t)2048 * 2560 * 2)); CK(hipMalloc(&dO, (size_t)M * 2048 * 4));
    CK(hipMalloc(&db, 4096 * 4));
    std::vector<half_t> h((size_t)M * 2560);
    uint64_t sd = 1;
    for (auto &v : h) { sd = sd * 6364136223846793005ull + 1442695040888963407ull; v = (half_t)(((float)((sd >> 33) & 0xFFFF) / 32768.f - 1.f) * 0.5f); }
    CK(hipMemcpy(dA, h.data(), (size_t)M * 2560 * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, h.data(), (size_t)2048 * 2560 * 2, hipMemcpyHostToDevice));
    CK(hipMemset(db, 0, 4096 * 4));
    qv_gemm_set_t256(2);
    const size_t NB = 2048;
    unsigned long long *dT, *dP;
    CK(hipMalloc(&dT, NB * 8 * 64 * 8));
    CK(hipMalloc(&dP, NB * 4 * 8));
    std::vector<unsigned long long> hT(NB * 8 * 64), hP(NB * 4);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto &sh : shapes) {
        GemmArgs g = {};
        g.A = dA; g.W = dW; g.bias = db; g.out = dO; g.M = M; g.N = sh.N; g.K = sh.K; g.lda = sh.K; g.ldw = sh.K; g.ldo = sh.ldo; g.alpha = sh.alpha;
        const int nblk = (sh.N / 256) * ((M + 255) / 256), nk = sh.K / 64;
        printf("%s M=%d: %d blocks, %d K-tiles\n", sh.name, M, nblk, nk);
        const char *names[] = {"full", "no MFMA", "no frag reads", "no MFMA, no frag reads", "no ds_write", "no loads", "no loads, no ds_write",
                               "MFMA only (no loads/writes/reads)", "barriers + epilogue only", "full K loop, no epilogue", "full, epilogue without global loads/stores", "K loop off, epilogue without global accesses", "K loop off, no epilogue", "full, all blocks load tile (0,0)", "no epilogue, all blocks load tile (0,0)", "full, stores then loads (not interleaved)", "full"};
        const int masks[] = {0, 1, 2, 3, 4, 8, 12, 14, 15, 32, 16, 15 + 16, 15 + 32, 64, 64 + 32, 256, 0};
        for (int v = 0; v < 17; ++v) {
            g.abl = masks[v];
            for (int i = 0; i < 3; ++i) launch_gemm(sh.epi, g, 0);
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 20; ++i) launch_gemm(sh.epi, g, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("    %-40s %7.2f us\n", names[v], ms * 1e3 / 20);
        }
        for (int rep = 0; rep < 3; ++rep) {
        g.abl = rep == 1 ? 256 : 0;
        if (rep == 2) CK(hipMemset(dA, 0, (size_t)M * 2560 * 2));
        printf("    -- traced launch: %s\n", rep == 0 ? "full" : rep == 1 ? "stores then loads (ablation 256)" : "full, A = 0");
        g.trace = nullptr; g.phase = dP;
        for (int i = 0; i < 5; ++i) launch_gemm(sh.epi, g, 0);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hP.data(), dP, hP.size() * 8, hipMemcpyDeviceToHost));
        {
            double pro = 0, loop = 0;
            for (int b = 0; b < nblk; ++b) {
                pro += (double)(hP[b * 4 + 1] - hP[b * 4]) / 100.0;
                loop += (double)(hP[b * 4 + 2] - hP[b * 4 + 1]) / 100.0;
            }
            printf("    block timeline WITHOUT per-phase stamps: prologue %.2f us, K loop %.2f us = %.3f us per K-tile\n", pro / nblk, loop / nblk, loop / nblk / nk);
        }
        g.trace = dT;
        CK(hipMemset(dT, 0, NB * 8 * 64 * 8));
        for (int i = 0; i < 5; ++i) launch_gemm(sh.epi, g, 0);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hT.data(), dT, hT.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hP.data(), dP, hP.size() * 8, hipMemcpyDeviceToHost));
        // block timeline (100 MHz wall clock -> us)
        unsigned long long t0 = ~0ull, t3 = 0;
        double pro = 0, loop = 0, epi = 0;
        for (int b = 0; b < nblk; ++b) {
            t0 = std::min(t0, hP[b * 4]);
            t3 = std::max(t3, hP[b * 4 + 2]);
            pro += (double)(hP[b * 4 + 1] - hP[b * 4]) / 100.0;
            loop += (double)(hP[b * 4 + 2] - hP[b * 4 + 1]) / 100.0;
        }
        double start_spread = 0;
        for (int b = 0; b < nblk; ++b) start_spread = std::max(start_spread, (double)(hP[b * 4] - t0) / 100.0);
        printf("    block timeline: prologue %.2f us, K loop %.2f us (mean over blocks); first entry -> last K-loop end %.2f us; entry spread %.2f us\n",
               pro / nblk, loop / nblk, (double)(t3 - t0) / 100.0, start_spread);
        printf("    K-tile cycles (wave 0, s_memtime):");
        for (int kt = 0; kt < nk && kt < 12; ++kt) {
            double sm = 0;
            for (int b = 0; b < nblk; ++b) sm += (double)(hT[((size_t)b * 8) * 64 + kt + 1] - hT[((size_t)b * 8) * 64 + kt]);
            printf(" %.0f", sm / nblk);
        }
        printf("\n");
        g.trace = nullptr; g.phase = nullptr;
        }
        CK(hipMemcpy(dA, h.data(), (size_t)M * 2560 * 2, hipMemcpyHostToDevice));
        g.abl = 0;
    }
    return 0;
}
