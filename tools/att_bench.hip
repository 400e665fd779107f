// att_bench.hip -- the attention kernels on their own: B utterances of T frames (default 64 x 126, the headline batch),
// every variant of launch_attention timed over back-to-back launches, and shader-clock stamps of one block of
// k_attention_short at its phase boundaries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DQV_ATT_STAMPS -I offline-tarteel_amd/csrc -I include \
//         tools/att_bench.hip -o tools/att_bench
#include "../offline-tarteel_amd/csrc/qv_layers.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 126, iters = argc > 3 ? atoi(argv[3]) : 200;
    const int t_pad = (T + 31) / 32 * 32, M = B * T, pos_ld = 17 * QV_D;
    std::vector<half_t> hqk((size_t)M * 2 * QV_D), hvt((size_t)B * QV_D * t_pad), hpos((size_t)(2 * T - 1) * pos_ld);
    srand(1);
    auto rnd = [] { return (half_t)((rand() % 2001 - 1000) / 1000.0f); };
    for (auto &x : hqk) x = rnd();
    for (auto &x : hvt) x = rnd();
    for (auto &x : hpos) x = rnd();
    std::vector<float> hb(QV_D, 0.01f);
    std::vector<int32_t> hlen(B, T), hoff(B);
    for (int b = 0; b < B; ++b) hoff[b] = b * T;
    half_t *qk, *vt, *pos, *out, *out2;
    float *bu, *bv;
    int32_t *len, *off;
    CK(hipMalloc(&qk, hqk.size() * 2)); CK(hipMalloc(&vt, hvt.size() * 2)); CK(hipMalloc(&pos, hpos.size() * 2));
    CK(hipMalloc(&out, (size_t)M * QV_D * 2)); CK(hipMalloc(&out2, (size_t)M * QV_D * 2));
    CK(hipMalloc(&bu, QV_D * 4)); CK(hipMalloc(&bv, QV_D * 4)); CK(hipMalloc(&len, B * 4)); CK(hipMalloc(&off, B * 4));
    CK(hipMemcpy(qk, hqk.data(), hqk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(vt, hvt.data(), hvt.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(pos, hpos.data(), hpos.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(bu, hb.data(), QV_D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bv, hb.data(), QV_D * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(len, hlen.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(off, hoff.data(), B * 4, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<half_t> ref((size_t)M * QV_D), got((size_t)M * QV_D);
    for (int variant : {0, 1, 2, 4, 3}) {      // 4 = k_attention_x (no loader waves, two independent blocks per CU)
        if (variant == 3 && T > 128) continue;
        qv_attention_set_variant(variant);
        half_t *o = variant == 0 ? out : out2;
        CK(hipMemsetAsync(o, 0, (size_t)M * QV_D * 2, s));
        for (int i = 0; i < 20; ++i) launch_attention(qk, vt, pos, pos_ld, bu, bv, len, off, o, T, T, t_pad, B, s);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) launch_attention(qk, vt, pos, pos_ld, bu, bv, len, off, o, T, T, t_pad, B, s);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy((variant == 0 ? ref : got).data(), o, (size_t)M * QV_D * 2, hipMemcpyDeviceToHost));
        double md = 0;
        if (variant) for (size_t i = 0; i < ref.size(); ++i) md = std::max(md, (double)fabsf((float)ref[i] - (float)got[i]));
        printf("variant %d: %.2f us per launch (back to back), max |diff| to variant 0 = %.3g\n", variant, ms * 1000 / iters, md);
    }
    if (T > 128) {
        // variant 0 (two heads per block) ran first and variant 1, 2 overwrote nothing of it: re-run variant 0 once for the stamps
        qv_attention_set_variant(0);
        launch_attention(qk, vt, pos, pos_ld, bu, bv, len, off, out, T, T, t_pad, B, s);
        CK(hipStreamSynchronize(s));
        static long long st[12][16][6];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_ws_stamp), sizeof(st)));
        const int n_kt = (T + 31) / 32 < 16 ? (T + 31) / 32 : 16;
        const long long t0 = st[0][0][0];
        printf("k_attention_ws<2,2,192> block (head pair 1, query group 1, utterance 17): clocks since consumer wave 0 reached its first barrier\n");
        for (int w : {0, 1, 4, 8, 9}) {
            printf("  %s wave %2d:", w < 8 ? "consumer" : "loader  ", w);
            for (int kt = 0; kt < n_kt; ++kt) {
                if (w < 8) printf(" | kt%d at-barrier %lld released %lld skewed %lld softmax %lld pv %lld", kt, st[w][kt][0] - t0, st[w][kt][1] - t0, st[w][kt][2] - t0, st[w][kt][3] - t0, st[w][kt][4] - t0);
                else printf(" | kt%d wait %lld landed %lld released %lld issued %lld", kt, st[w][kt][0] - t0, st[w][kt][1] - t0, st[w][kt][2] - t0, st[w][kt][3] - t0);
                if (kt == 3) { printf(" ..."); kt = n_kt - 3; }
            }
            printf("\n");
        }
    }
    if (T <= 128) {
        long long st[4][8];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_att_stamp), sizeof(st)));
        for (int w = 0; w < 4; ++w) {
            printf("k_attention_short block (head 3, utterance 17) wave %d, shader clocks since entry:", w);
            for (int i = 1; i < 7; ++i) printf(" %lld", st[w][i] - st[w][0]);
            printf("   [loads issued, staged, products done, skewed, softmax, end]\n");
        }
    }
    return 0;
}
