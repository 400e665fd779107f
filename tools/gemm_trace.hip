// gemm_trace.hip -- dev tool: s_memtime stamps inside k_gemm's K loop (per wave role), FFN-up / FFN-down shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DQV_GEMM_TRACE tools/gemm_trace.hip -o tools/gemm_trace
//   QVERSE_GEMM_LD=1 tools/gemm_trace
// Prints, averaged over all blocks: for a consumer wave the time from barrier release to its arrival at
// the next barrier (= its ds_read + MFMA chain) and the time it then waits; for a loader wave the time
// until its stage is written, until it has re-armed its loads, and its wait.
#include "../offline-tarteel_amd/csrc/qv_gemm.hip"
#include "../offline-tarteel_amd/csrc/qv_gemm256.hip"

#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
    qv_gemm_set_t256(0);
    const int M = 8064;
    struct Sh { const char *name; int epi, N, K, ldo; float alpha; } shapes[] = {
        {"ff_up   N2048 K512 ", EPI_F16_SWISH, 2048, 512, 2048, 1.f},
        {"ff_down N512 K2048 ", EPI_RESID, 512, 2048, 512, 0.5f},
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
    const size_t TR = (size_t)1024 * 8 * 64 * 4;
    unsigned long long *dT;
    CK(hipMalloc(&dT, TR * 8));
    std::vector<unsigned long long> hT(TR);
    for (auto &sh : shapes) {
        GemmArgs g = {};
        g.A = dA; g.W = dW; g.bias = db; g.out = dO; g.M = M; g.N = sh.N; g.K = sh.K; g.lda = sh.K; g.ldw = sh.K; g.ldo = sh.ldo; g.alpha = sh.alpha;
        g.trace = nullptr;
        {
            // ablations, untraced: event-timed averages of 20 launches
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const char *names[] = {"full", "no MFMA", "no frag reads", "no MFMA, no frag reads", "no ds_write", "no loads", "no loads, no ds_write",
                                   "loaders idle + no frag reads (MFMA only)", "everything off (barriers + epilogue)",
                                   "K loop off, epilogue without global loads/stores", "K loop off, no epilogue", "full K loop, no epilogue", "full, epilogue without global loads/stores", "full, stores then loads (not interleaved)", "full"};
            const int masks[] = {0, 1, 2, 3, 4, 8, 12, 14, 15, 15 + 16, 15 + 32, 32, 16, 256, 0};
            for (int v = 0; v < 15; ++v) {
                g.abl = masks[v];
                for (int i = 0; i < 3; ++i) launch_gemm(sh.epi, g, 0);
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 20; ++i) launch_gemm(sh.epi, g, 0);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("    %-45s %7.2f us\n", names[v], ms * 1e3 / 20);
            }
            g.abl = 0;
        }
        {
            // block-level timeline (100 MHz wall clock, comparable across CUs): when blocks start, how long each phase lasts
            unsigned long long *dP;
            const int nblk = (sh.N / 128) * ((M + 127) / 128);
            CK(hipMalloc(&dP, (size_t)nblk * 4 * 8));
            CK(hipMemset(dP, 0, (size_t)nblk * 4 * 8));
            for (int i = 0; i < 3; ++i) launch_gemm(sh.epi, g, 0);
            CK(hipDeviceSynchronize());
            g.phase = dP;
            launch_gemm(sh.epi, g, 0);
            CK(hipDeviceSynchronize());
            g.phase = nullptr;
            std::vector<unsigned long long> hP((size_t)nblk * 4);
            CK(hipMemcpy(hP.data(), dP, hP.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int b = 0; b < nblk; ++b) { t0 = std::min(t0, hP[b * 4]); t1 = std::max(t1, hP[b * 4 + 3]); }
            double pro = 0, kl = 0, ep = 0;
            std::vector<double> starts;
            for (int b = 0; b < nblk; ++b) {
                pro += (double)(hP[b * 4 + 1] - hP[b * 4]); kl += (double)(hP[b * 4 + 2] - hP[b * 4 + 1]); ep += (double)(hP[b * 4 + 3] - hP[b * 4 + 2]);
                starts.push_back((double)(hP[b * 4] - t0) / 100.0);
            }
            std::sort(starts.begin(), starts.end());
            printf("    timeline: kernel span %.2f us (first block entry -> last block exit); per block: entry->first barrier %.2f us, K loop %.2f us, epilogue %.2f us\n",
                   (double)(t1 - t0) / 100.0, pro / nblk / 100.0, kl / nblk / 100.0, ep / nblk / 100.0);
            printf("    block entry times (us after the first): p10 %.2f  p25 %.2f  p50 %.2f  p75 %.2f  p90 %.2f  max %.2f\n", starts[nblk / 10], starts[nblk / 4],
                   starts[nblk / 2], starts[3 * nblk / 4], starts[9 * nblk / 10], starts.back());
            CK(hipFree(dP));
        }
        for (int i = 0; i < 3; ++i) launch_gemm(sh.epi, g, 0);
        CK(hipMemset(dT, 0, TR * 8));
        g.trace = dT;
        launch_gemm(sh.epi, g, 0);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hT.data(), dT, TR * 8, hipMemcpyDeviceToHost));
        const int nk = sh.K / 64, nblk = (sh.N / 128) * ((M + 127) / 128);
        // consumer wave 0, loader wave 4
        double c_work = 0, c_wait = 0, l_put = 0, l_arm = 0, l_wait = 0, step = 0; long n = 0, ns = 0;
        for (int b = 0; b < nblk && b < 1024; ++b) {
            auto T = [&](int wave, int kt, int slot) { return (double)hT[(((size_t)b * 8 + wave) * 64 + kt) * 4 + slot]; };
            for (int kt = 1; kt + 1 < nk; ++kt) {
                if (!T(0, kt, 0) || !T(4, kt, 0)) continue;
                c_work += T(0, kt + 1, 2) - T(0, kt, 0);      // released at barrier kt -> arrives at barrier kt + 1
                c_wait += T(0, kt + 1, 0) - T(0, kt + 1, 2);
                l_put += T(4, kt, 1) - T(4, kt, 0);
                l_arm += T(4, kt, 2) - T(4, kt, 1);
                if (kt + 1 < nk) l_wait += T(4, kt + 1, 0) - T(4, kt, 2);
                step += T(0, kt + 1, 0) - T(0, kt, 0);
                ++n;
            }
        }
        printf("%s blocks %d nk %d | K-step %.0f clk | consumer: work %.0f wait %.0f | loader: wait-loads+write %.0f re-arm %.0f wait %.0f  (s_memtime ticks, avg over %ld steps)\n",
               sh.name, nblk, nk, step / n, c_work / n, c_wait / n, l_put / n, l_arm / n, l_wait / n, n);
        // whole-block span: first stamp to last stamp
        double span = 0; int nb = 0;
        for (int b = 0; b < nblk && b < 1024; ++b) {
            double t0 = (double)hT[(((size_t)b * 8 + 0) * 64 + 0) * 4 + 2], t1 = (double)hT[(((size_t)b * 8 + 0) * 64 + nk - 1) * 4 + 0];
            if (t0 && t1) { span += t1 - t0; ++nb; }
        }
        printf("    consumer wave 0: first barrier arrival -> last barrier release: %.0f ticks avg\n", span / nb);
    }
    return 0;
}
