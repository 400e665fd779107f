// qv_postlogits.hip -- everything after the log-probs, on device:
//   argmax -> CTC collapse -> piece expansion + normalisation  (c2c-direct/run.py:187-204)
//   trigram-IDF candidates, fragment scores, span pass          (shared/quran_db.py:173-186,211-237,244-371)
//   search / pass-3 / candidate assembly                        (quran_db.py:92-99; c2c-direct/run.py:251-311)
//   CTC alpha-recursion rerank + decision                       (c2c-direct/run.py:314-380; mixed/run.py:96-133)
//
// Integer/byte work bound by LDS + VALU, not MFMA: the unit of work is a bit-parallel LCS
// (Indel distance) between the transcript and a verse text or text window.  Mapping:
//   * "one text per lane" kernels (pass 3, span pass): pattern = transcript bit-vector (<=16 x
//     64-bit words in VGPRs), each lane streams its own verse/span text.
//   * "one text per wave" kernel (fragment score): lanes = sliding windows of partial_ratio,
//     pattern = the shorter string's match masks, lane 0 additionally does the full-string LCS.
//   * CTC: one wave per (utterance, candidate), 2L+1 states laid contiguously across lanes,
//     two DPP-style shuffles per frame.
// All double arithmetic that feeds comparisons uses explicit _rn intrinsics (no FMA contraction)
// so scores are bit-identical to the reference's Python floats.

#include "qv_common.h"
#include "qv_kernels.h"   // qv_kernel_variant

#include <math.h>

#include <algorithm>
#include <type_traits>

namespace {

// ------------------------------------------------------------------ small helpers ------

__device__ __forceinline__ double ratio_from(int lcs, int la, int lb) {
    int tot = la + lb;
    if (tot == 0) return 1.0;
    return __dsub_rn(1.0, __ddiv_rn((double)(tot - 2 * lcs), (double)tot));
}

__device__ __forceinline__ int wave_rank(bool pred, int lane, int &total) {
    unsigned long long m = __ballot(pred);
    total = __popcll(m);
    return __popcll(m & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// (score desc, key asc) ordering used by every stable "sorted(..., reverse=True)" emulation
__device__ __forceinline__ bool better(double s, unsigned long long k, double s2, unsigned long long k2) {
    return s > s2 || (s == s2 && k < k2);
}

__device__ __forceinline__ void wave_best(double &s, unsigned long long &k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double s2 = __shfl_xor(s, o);
        unsigned long long k2 = __shfl_xor(k, o);
        if (better(s2, k2, s, k)) { s = s2; k = k2; }
    }
}

// block-wide best over 256 threads; result valid in all threads.  sh_s/sh_k: [4+1] scratch.
__device__ __forceinline__ void block_best(double &s, unsigned long long &k, double *sh_s, unsigned long long *sh_k) {
    wave_best(s, k);
    int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh_s[w] = s; sh_k[w] = k; }
    __syncthreads();
    s = sh_s[0]; k = sh_k[0];
    for (int i = 1; i < nw; ++i)
        if (better(sh_s[i], sh_k[i], s, k)) { s = sh_s[i]; k = sh_k[i]; }
}

// best non-negative entry among the positions p = tid, tid + 256, ... that this thread owns.
// Top-k selection keeps each thread's best cached: after a round only the owner of the winner
// rescans its ~25 entries, so a round costs one block reduction instead of a full scan.
__device__ __forceinline__ void local_best(const double *sc, int n, int tid, double &s, unsigned long long &k) {
    s = -1.0;
    k = ~0ull;
    for (int p = tid; p < n; p += 256) {
        double x = sc[p];
        if (x >= 0.0 && better(x, (unsigned long long)p, s, k)) { s = x; k = p; }
    }
}

// Top-K of sc[0..n) (entries < 0 excluded) in the order of a stable descending sort, i.e.
// (score desc, position asc), for a 256-thread block.  MSB-first radix select on the
// order-preserving bit pattern of the non-negative doubles finds the K-th largest value in 8
// histogram passes; ties at that value are taken in position order; the <= K survivors are
// ordered by rank counting.  scratch: u32[256 + 8], sel_p: i32[K], sel_s: f64[K] (LDS).
// out_p / out_s may be global.  Returns the number selected (same in every thread).
__device__ int block_topk_select(const double *sc, int n, int K, uint32_t *scratch, int32_t *sel_p, double *sel_s,
                                 int32_t *out_p, double *out_s) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *hist = scratch, *misc = scratch + 256;  // misc: [0] digit, [1] need, [2] count, [3..6] wave sums
    auto key_of = [&](int p) -> unsigned long long {
        double x = sc[p];
        return x >= 0.0 ? (unsigned long long)__double_as_longlong(x) + 1ull : 0ull;
    };
    // number of valid entries
    int cnt = 0;
    for (int p = tid; p < n; p += 256) cnt += sc[p] >= 0.0;
    cnt = wave_sum_i(cnt);
    if (lane == 0) misc[3 + wave] = cnt;
    if (tid == 0) misc[2] = 0;
    __syncthreads();
    const int nvalid = misc[3] + misc[4] + misc[5] + misc[6];
    if (K > nvalid) K = nvalid;
    if (K <= 0) return 0;
    unsigned long long prefix = 0, mask = 0;
    int need = K;
    for (int shift = 56; shift >= 0; shift -= 8) {
        __syncthreads();
        hist[tid] = 0;
        __syncthreads();
        for (int p = tid; p < n; p += 256) {
            unsigned long long k = key_of(p);
            if (k != 0 && (k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1u);
        }
        __syncthreads();
        if (wave == 0) {
            // bins 255..0: lane l owns bins 4l..4l+3; suffix sums from the top
            uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            uint32_t tot = h0 + h1 + h2 + h3, suf = tot;  // inclusive suffix over lanes >= l
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                uint32_t x = __shfl_down(suf, o);
                if (lane + o < 64) suf += x;
            }
            uint32_t above = suf - tot;  // elements in bins of higher lanes
            bool here = above < (uint32_t)need && suf >= (uint32_t)need;
            if (here) {
                uint32_t c = above;
                int d;
                if (c + h3 >= (uint32_t)need) d = 3;
                else { c += h3; if (c + h2 >= (uint32_t)need) d = 2; else { c += h2; if (c + h1 >= (uint32_t)need) d = 1; else { c += h1; d = 0; } } }
                misc[0] = 4 * lane + d;
                misc[1] = need - (int)c;
            }
        }
        __syncthreads();
        prefix |= (unsigned long long)misc[0] << shift;
        mask |= 0xFFull << shift;
        need = (int)misc[1];
    }
    const unsigned long long T = prefix;  // key of the K-th largest; `need` of the == T entries are taken
    // entries above T (unordered) ...
    for (int p = tid; p < n; p += 256) {
        unsigned long long k = key_of(p);
        if (k > T) { int slot = atomicAdd(&misc[2], 1u); sel_p[slot] = p; sel_s[slot] = sc[p]; }
    }
    // ... and the first `need` entries equal to T, in position order
    int taken = 0;
    for (int base = 0; base < n && taken < need; base += 256) {
        int p = base + tid;
        bool eq = p < n && key_of(p) == T;
        unsigned long long bal = __ballot(eq);
        int r = __popcll(bal & ((1ull << lane) - 1ull)), tot = __popcll(bal);
        __syncthreads();
        if (lane == 0) misc[3 + wave] = tot;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; ++w) before += misc[3 + w];
        int all = misc[3] + misc[4] + misc[5] + misc[6];
        if (eq && taken + before + r < need) { int slot = atomicAdd(&misc[2], 1u); sel_p[slot] = p; sel_s[slot] = sc[p]; }
        taken += all;
    }
    __syncthreads();
    // order the K survivors
    for (int i = tid; i < K; i += 256) {
        double si = sel_s[i];
        int pi = sel_p[i], rank = 0;
        for (int j = 0; j < K; ++j) rank += better(sel_s[j], (unsigned long long)sel_p[j], si, (unsigned long long)pi);
        out_p[rank] = pi;
        out_s[rank] = si;
    }
    __syncthreads();
    return K;
}

#ifndef QV_LCS_ADDC
#define QV_LCS_ADDC 1   // 0: the compiler's own carry chain (cross-check builds)
#endif
// ------------------------------------------------------------------ bit-parallel LCS ---
// Hyyro/Crochemore: V all ones; per text char U = V & M; V = (V + U) | (V & ~M).
// pm: match masks [sym][stride] (u64), W words used; text codes >= QV_NSYM match nothing.
// lcs_chunk advances the recurrence over the first cnt (<= 8) codes packed in `chunk`; lcs_feed continues it over n
// more text codes (V carries the state); lcs_count reads the LCS length of everything fed so far: LCS(pattern, text[:j]) is
// available at every j of one walk over the text.
// 64-bit a + b + carry-in -> sum, carry-out, with the carry held as a LANE MASK in an SGPR pair (v_addc_co_u32's VOP3 form takes any
// pair as carry source and destination): two instructions per word.  The generic lowering of __builtin_addcll costs two 64-bit
// adds, two 64-bit compares and the selects that turn them into a carry value -- the window scans, the full-string pass and the
// span pass are bound by VALU issue (profiles/r06_c_pmc_post.txt: 202 M / 82 M / 118 M VALU instructions per launch), and the carry
// was half of a word-step's instructions.
__device__ __forceinline__ uint64_t addc64(uint64_t a, uint64_t b, unsigned long long &carry) {
    uint32_t lo, hi;
    asm("v_addc_co_u32 %0, %2, %3, %4, %2\n\tv_addc_co_u32 %1, %2, %5, %6, %2"
        : "=&v"(lo), "=&v"(hi), "+s"(carry)
        : "v"((uint32_t)a), "v"((uint32_t)b), "v"((uint32_t)(a >> 32)), "v"((uint32_t)(b >> 32)));
    return ((uint64_t)hi << 32) | lo;
}

// ZROW: the mask table has an all-zero row at index QV_NSYM (the LDS copies: load_pm_lds, k_frag's slice): codes past cnt
// and codes outside the alphabet select THAT row, instead of a select per mask word (2 of a word-step's ~10 instructions).
template <int W, bool ZROW = false>
__device__ __forceinline__ void lcs_chunk(uint64_t (&V)[W], const uint64_t *__restrict__ pm, int stride, uint64_t chunk, int cnt) {
    // The match masks of CH consecutive codes are requested TOGETHER, before the dependent add chain
    // of those steps: the recurrence is one serial chain per lane, and a mask load inside every step
    // (L1/L2 or LDS latency each) used to be most of a step's time.  Codes past cnt
    // and codes outside the alphabet get an all-zero mask, which leaves V unchanged.
    constexpr int CH = W <= 4 ? 8 : (W <= 8 ? 4 : 1);   // (wider patterns: fewer masks in flight -- mk[CH][W] sets the kernels' VGPR count, i.e. how many waves a SIMD interleaves)
#pragma unroll
    for (int g0 = 0; g0 < 8; g0 += CH) {
        if (g0 >= cnt) break;
        uint64_t mk[CH][W];
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            int c = (int)((chunk >> (8 * (g0 + e))) & 0xFF);
            const bool valid = g0 + e < cnt && c < QV_NSYM;
            const uint64_t *M = pm + (size_t)(valid ? c : (ZROW ? QV_NSYM : 0)) * stride;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                uint64_t x = M[w];
                mk[e][w] = (ZROW || valid) ? x : 0ull;
            }
        }
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            unsigned long long carry = 0;   // lane mask of the carries (addc64)
#pragma unroll
            for (int w = 0; w < W; ++w) {
                uint64_t v = V[w], mm = mk[e][w];
                // v + (v & mm) + carry with the carry chained through the words
                uint64_t s2 = QV_LCS_ADDC ? addc64(v, v & mm, carry) : __builtin_addcll(v, v & mm, carry, &carry);
                V[w] = s2 | (v & ~mm);
            }
        }
    }
}

typedef uint64_t __attribute__((aligned(1))) u64_unaligned;

template <int W, bool ZROW = false>
__device__ __forceinline__ void lcs_feed(uint64_t (&V)[W], const uint64_t *__restrict__ pm, int stride,
                                         const uint8_t *__restrict__ text, int n) {
    // the text is fetched 8 codes per (possibly unaligned) load: one memory access per 8 steps
    // of the recurrence instead of one per step; every text buffer is padded by >= 8 bytes
    // ... and the NEXT 8 codes are requested before this chunk's steps run: the lane's chain used to start every chunk
    // with an exposed global-load latency, and a kernel lasts as long as its longest text (1,100+ codes = 140 chunks)
    uint64_t next = n > 0 ? *(const u64_unaligned *)text : 0ull;
    for (int j0 = 0; j0 < n; j0 += 8) {
        const uint64_t chunk = next;
        if (j0 + 8 < n) next = *(const u64_unaligned *)(text + j0 + 8);
        lcs_chunk<W, ZROW>(V, pm, stride, chunk, n - j0 < 8 ? n - j0 : 8);
    }
}

template <int W>
__device__ __forceinline__ int lcs_count(const uint64_t (&V)[W], int m) {
    int zeros = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        uint64_t z = ~V[w];
        int lo = w * 64;
        if (m < lo + 64) z &= (m > lo) ? ((1ull << (m - lo)) - 1ull) : 0ull;
        zeros += __popcll(z);
    }
    return zeros;
}

template <int W, bool ZROW = false>
__device__ __forceinline__ int lcs_core(const uint64_t *__restrict__ pm, int stride, const uint8_t *__restrict__ text,
                                        int n, int m) {
    uint64_t V[W];
#pragma unroll
    for (int w = 0; w < W; ++w) V[w] = ~0ull;
    lcs_feed<W, ZROW>(V, pm, stride, text, n);
    return lcs_count<W>(V, m);
}

template <bool ZROW = false>
__device__ __forceinline__ int lcs_dispatch(int W, const uint64_t *pm, int stride, const uint8_t *text, int n, int m) {
    if (n <= 0 || m <= 0) return 0;
    if (W <= 1) return lcs_core<1, ZROW>(pm, stride, text, n, m);
    if (W <= 2) return lcs_core<2, ZROW>(pm, stride, text, n, m);
    if (W <= 3) return lcs_core<3, ZROW>(pm, stride, text, n, m);
    if (W <= 4) return lcs_core<4, ZROW>(pm, stride, text, n, m);
    if (W <= 6) return lcs_core<6, ZROW>(pm, stride, text, n, m);
    if (W <= 8) return lcs_core<8, ZROW>(pm, stride, text, n, m);
    if (W <= 11) return lcs_core<11, ZROW>(pm, stride, text, n, m);
    return lcs_core<16, ZROW>(pm, stride, text, n, m);
}

// ---- the same recurrence with the pattern's words spread over G neighbouring lanes (G = 4, 8, 16; lane w of the group
// owns word w) as a skewed wavefront: at step t lane w applies text code j = t - w, with the carry lane w - 1 produced for
// that code one step earlier (one DPP row shift).  A lane's serial chain is then one 64-bit add per text code instead of
// W of them: the span pass and the window scans last as long as their LONGEST lane, and for transcripts of several hundred
// characters (W = 4..7) that lane used to walk 1,000+ codes x W words.  Same integer result as lcs_core.
// All G lanes pass the same text / n / m / W; the return value is valid in every lane of the group.
template <int G>
__device__ __forceinline__ int lcs_systolic(const uint64_t *__restrict__ pm, int stride, const uint8_t *__restrict__ text, int n,
                                            int m, int W, int w) {
    uint64_t V = ~0ull;
    unsigned cout = 0;                      // the carry this lane produced in its last step
    const bool act = w < W;
    const int total = n + G - 1;            // steps t = 0 .. total - 1; lane w is at code j = t - w
    for (int t0 = 0; t0 < total; t0 += 8) {
        const int j0 = t0 - w;
        uint64_t chunk = 0;                 // the lane's next 8 codes (text buffers are padded by >= 8 bytes)
        if (j0 < n && j0 > -8) chunk = j0 >= 0 ? *(const u64_unaligned *)(text + j0) : (*(const u64_unaligned *)text << (8 * -j0));
        uint64_t mk[8];                     // their match-mask words, requested together in front of the dependent chain
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = j0 + e, c = (int)((chunk >> (8 * e)) & 0xFF);
            const bool valid = act && j >= 0 && j < n && c < QV_NSYM;
            const uint64_t x = pm[(size_t)(valid ? c : 0) * stride + (act ? w : 0)];
            mk[e] = valid ? x : 0ull;       // a zero mask leaves V alone and produces no carry
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            unsigned cin = (unsigned)__builtin_amdgcn_update_dpp(0, (int)cout, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
            if (w == 0) cin = 0;
            unsigned long long carry;
            const uint64_t s2 = __builtin_addcll(V, V & mk[e], (unsigned long long)cin, &carry);
            V = s2 | (V & ~mk[e]);
            cout = (unsigned)carry;
        }
    }
    uint64_t z = ~V;
    const int lo = w * 64;
    if (m < lo + 64) z &= (m > lo) ? ((1ull << (m - lo)) - 1ull) : 0ull;
    int cnt = act ? __popcll(z) : 0;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    return cnt;
}

// stage the transcript's match masks (both the spaced and the spaceless pattern, 10 KB) in LDS
// LDS row stride of the match masks (u64 units).  With the global stride QV_MAXW = 16 (128 B) every
// symbol's word w sat in the same two banks, so the 64 lanes of a wave - each on a different
// symbol - serialised their mask reads; the padded stride spreads the rows over the banks.
#ifndef QV_PMS
#define QV_PMS (QV_MAXW + 2)   // 144 B rows: 16-byte aligned for ds_read_b128, 16 distinct bank offsets
#endif
#define QV_PM_ROWS (QV_NSYM + 1)   // rows per pattern of an LDS copy: the alphabet + one all-zero row (lcs_chunk<W, true>)
__device__ __forceinline__ void load_pm_lds(uint64_t *spm, const uint64_t *gpm) {
    for (int i = threadIdx.x; i < 2 * QV_NSYM * QV_MAXW; i += blockDim.x) {
        const int row = i / QV_MAXW, p = row / QV_NSYM;
        spm[(p * QV_PM_ROWS + row - p * QV_NSYM) * QV_PMS + (i % QV_MAXW)] = gpm[i];
    }
    for (int i = threadIdx.x; i < 2 * QV_PMS; i += blockDim.x) spm[((i / QV_PMS) * QV_PM_ROWS + QV_NSYM) * QV_PMS + (i % QV_PMS)] = 0ull;
    __syncthreads();
}

// ------------------------------------------------------------------ 1. argmax + decode --
// One 1024-thread block per utterance (three launches in the first version: reset, argmax per frame, decode):
//   all sixteen waves: per-frame argmax with numpy semantics (first maximum), frame ids kept in LDS;
//   wave 0 alone:   CTC collapse, piece expansion into normalised codes, whitespace collapse + strip, spaceless copy,
//                   match masks (c2c-direct/run.py:187-204, shared/normalizer.py) -- a serial, 64-lane job.
// Within that single wave the LDS / global hand-offs between lanes only need the wave's own memory operations to have
// completed (QV_WSYNC: workgroup-scope release / acquire fences around a wave barrier), not a block barrier.
#define QV_WSYNC()                                                  \
    do {                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      \
        __builtin_amdgcn_wave_barrier();                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      \
    } while (0)
#define QV_TCAP 770   // >= t_cap of any engine (qv_create refuses more than 768 + 2 frames)

#define DEC_WAVES 16
__global__ __launch_bounds__(64 * DEC_WAVES) void k_decode(QvTables tab, QvWork wk, const float *__restrict__ lp, int t_max,
                                                          const int32_t *__restrict__ t_dev) {
    __shared__ uint8_t raw[QV_RAW_CAP];
    __shared__ unsigned long long pm[2][QV_NSYM][QV_MAXW];
    __shared__ int16_t s_fid[QV_TCAP];
    __shared__ int32_t s_tok[QV_TCAP];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (b == 0 && tid == 0) { *wk.n_fail = 0; wk.frag_ctr[0] = 0; wk.frag_ctr[1] = 0; wk.frag_ctr[2] = 0; }
    QvUtt &u = wk.utt[b];
    const int T = t_dev[b];
    // a wave takes every 16th frame; the 17 loads of a row are all requested before the first comparison (a frame's
    // argmax is one memory latency, not seventeen)
    for (int t = wave; t < T; t += DEC_WAVES) {
        const float *row = lp + ((size_t)b * t_max + t) * QV_VOCAB;
        float x[17];
#pragma unroll
        for (int i = 0; i < 17; ++i) x[i] = lane + 64 * i < QV_VOCAB ? row[lane + 64 * i] : -INFINITY;
        float best = x[0];
        int bi = lane;
#pragma unroll
        for (int i = 1; i < 17; ++i)
            if (x[i] > best) { best = x[i]; bi = lane + 64 * i; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float b2 = __shfl_xor(best, o);
            int i2 = __shfl_xor(bi, o);
            if (b2 > best || (b2 == best && i2 < bi)) { best = b2; bi = i2; }
        }
        if (lane == 0) s_fid[t] = (int16_t)bi;
    }
    __syncthreads();
    if (wave != 0) return;
    int32_t *greedy = wk.greedy + (size_t)b * wk.t_cap;
    int n_tok = 0;
    for (int base = 0; base < T; base += 64) {
        int t = base + lane;
        int id = -1;
        bool keep = false;
        if (t < T) {
            id = s_fid[t];
            int prev = t > 0 ? s_fid[t - 1] : -1;
            keep = id != prev && id != QV_BLANK;
        }
        int tot, r = wave_rank(keep, lane, tot);
        if (keep) { s_tok[n_tok + r] = id; greedy[n_tok + r] = id; }
        n_tok += tot;
    }
    for (int t = n_tok + lane; t < wk.t_cap; t += 64) greedy[t] = -1;
    QV_WSYNC();
    // piece expansion
    int n_raw = 0;
    bool trunc = false;
    for (int base = 0; base < n_tok; base += 64) {
        int k = base + lane;
        int id = k < n_tok ? s_tok[k] : -1;
        int len = id >= 0 ? (int)(tab.piece_off[id + 1] - tab.piece_off[id]) : 0;
        int incl = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int x = __shfl_up(incl, o);
            if (lane >= o) incl += x;
        }
        int off = n_raw + incl - len;
        if (off + len > QV_RAW_CAP) { trunc = true; len = 0; }
        const uint8_t *src = tab.piece_codes + (id >= 0 ? tab.piece_off[id] : 0);
        for (int i = 0; i < len; ++i) raw[off + i] = src[i];
        n_raw += __shfl(incl, 63);
    }
    trunc = __any(trunc);
    if (n_raw > QV_RAW_CAP) n_raw = QV_RAW_CAP;
    QV_WSYNC();
    int last_ns = -1;
    for (int i = lane; i < n_raw; i += 64)
        if (raw[i] != 0) last_ns = i;
    last_ns = wave_max_i(last_ns);
    uint8_t *q = wk.q + (size_t)b * QV_MAXQ, *qs = wk.qs + (size_t)b * QV_MAXQ;
    // the collapsed codes go to HBM (every later kernel reads them there) and, for the match masks below, stay in
    // registers of the lane that produced them
    unsigned long long *pmf = &pm[0][0][0];
    for (int i = lane; i < 2 * QV_NSYM * QV_MAXW; i += 64) pmf[i] = 0ull;
    QV_WSYNC();
    int qn = 0, qsn = 0, spaces = 0;
    for (int base = 0; base < n_raw; base += 64) {
        int i = base + lane;
        int c = i < n_raw ? raw[i] : 0;
        bool keep = i < n_raw && (c != 0 ? true : (i > 0 && raw[i - 1] != 0 && i < last_ns));
        int tot, r = wave_rank(keep, lane, tot);
        if (keep && qn + r < QV_MAXQ) {
            q[qn + r] = (uint8_t)c;
            if (c < QV_NSYM) atomicOr(&pm[0][c][(qn + r) >> 6], 1ull << ((qn + r) & 63));
        }
        bool ks = keep && c != 0;
        int tots, rs = wave_rank(ks, lane, tots);
        if (ks && qsn + rs < QV_MAXQ) {
            qs[qsn + rs] = (uint8_t)c;
            if (c < QV_NSYM) atomicOr(&pm[1][c][(qsn + rs) >> 6], 1ull << ((qsn + rs) & 63));
        }
        spaces += tot - tots;
        qn += tot;
        qsn += tots;
    }
    if (qn > QV_MAXQ) trunc = true;
    int flags = 0;
    if (trunc) { flags |= QV_FLAG_TRANSCRIPT_TRUNCATED; qn = 0; qsn = 0; }
    else if (qn == 0) flags |= QV_FLAG_EMPTY_TRANSCRIPT;
    QV_WSYNC();
    uint64_t *gpm = wk.pm + (size_t)b * 2 * QV_NSYM * QV_MAXW;
    for (int i = lane; i < 2 * QV_NSYM * QV_MAXW; i += 64) gpm[i] = trunc ? 0ull : pmf[i];
    if (lane == 0) {
        u.t_frames = T;
        u.n_tok = n_tok;
        u.q_len = qn;
        u.qs_len = qsn;
        u.q_words = qn > 0 ? spaces + 1 : 0;
        u.flags = flags;
        u.base_start = -1; u.base_span = 0; u.base_score = 0.0;
        u.use_ctc = 0; u.n_cand = 0; u.win = -1; u.win_norm = 0.f;
        u.n_cand1 = 0; u.full_scan = 0; u.n_runners = 0; u.n_surah20 = 0; u.force_full = 0; u.hint_n = 0;
        u.best1_idx = -1; u.best1_score = 0.0;
    }
}

// debug path: transcript codes already in wk.q (uploaded by the host); build the rest.
__global__ __launch_bounds__(64) void k_prepare_codes(QvWork wk, int n) {
    __shared__ unsigned long long pm[2][QV_NSYM][QV_MAXW];
    int lane = threadIdx.x;
    QvUtt &u = wk.utt[0];
    uint8_t *q = wk.q, *qs = wk.qs;
    int qsn = 0, spaces = 0;
    for (int base = 0; base < n; base += 64) {
        int i = base + lane;
        int c = i < n ? q[i] : 0;
        bool ks = i < n && c != 0;
        int tot, r = wave_rank(ks, lane, tot);
        if (ks) qs[qsn + r] = (uint8_t)c;
        qsn += tot;
        int ts;
        wave_rank(i < n && c == 0, lane, ts);
        spaces += ts;
    }
    unsigned long long *pmf = &pm[0][0][0];
    for (int i = lane; i < 2 * QV_NSYM * QV_MAXW; i += 64) pmf[i] = 0ull;
    __syncthreads();
    for (int i = lane; i < n; i += 64) { int c = q[i]; if (c < QV_NSYM) atomicOr(&pm[0][c][i >> 6], 1ull << (i & 63)); }
    for (int i = lane; i < qsn; i += 64) { int c = qs[i]; if (c < QV_NSYM) atomicOr(&pm[1][c][i >> 6], 1ull << (i & 63)); }
    __syncthreads();
    for (int i = lane; i < 2 * QV_NSYM * QV_MAXW; i += 64) wk.pm[i] = pmf[i];
    if (lane == 0) {
        u.t_frames = 0; u.n_tok = 0; u.q_len = n; u.qs_len = qsn; u.q_words = n > 0 ? spaces + 1 : 0;
        u.flags = n == 0 ? QV_FLAG_EMPTY_TRANSCRIPT : 0;
        u.base_start = -1; u.base_span = 0; u.base_score = 0.0;
        u.use_ctc = 0; u.n_cand = 0; u.win = -1; u.win_norm = 0.f;
        u.n_cand1 = 0; u.full_scan = 0; u.n_runners = 0; u.n_surah20 = 0; u.best1_idx = -1; u.best1_score = 0.0;
        u.force_full = 0; u.hint_n = 0;
    }
}

// ------------------------------------------------------------------ 3. trigram top-50 --
// CPython set[int] iteration order of ints inserted one by one (Objects/setobject.c: linear
// probes 9, perturb shift 5, x4 growth at fill*5 >= mask*3); match_verse iterates
// set(top-50) (quran_db.py:279-288) and sorts stably, so exact ties resolve in this order.
__device__ int pyset_order(const int32_t *vals, int n, int32_t *out, int32_t *tabA, int32_t *tabB) {
    int mask = 7;
    int32_t *tab = tabA;
    for (int i = 0; i <= mask; ++i) tab[i] = -1;
    int fill = 0;
    for (int k = 0; k < n; ++k) {
        unsigned long long h = (unsigned long long)vals[k], perturb = h;
        int i = (int)(h & (unsigned)mask);
        int found = 0;
        for (;;) {
            int probes = (i + 9 <= mask) ? 9 : 0;
            int e = i;
            do {
                if (tab[e] < 0) { tab[e] = vals[k]; ++fill; found = 1; break; }
                if (tab[e] == vals[k]) { found = 2; break; }
                ++e;
            } while (probes--);
            if (found) break;
            perturb >>= 5;
            i = (int)(((unsigned long long)i * 5ull + 1ull + perturb) & (unsigned)mask);
        }
        if (found == 1 && fill * 5 >= mask * 3) {
            int minused = fill * 4, ns = 8;
            while (ns <= minused) ns <<= 1;
            int32_t *nt = (tab == tabA) ? tabB : tabA;
            for (int z = 0; z < ns; ++z) nt[z] = -1;
            int nmask = ns - 1;
            for (int z = 0; z <= mask; ++z) {
                if (tab[z] < 0) continue;
                unsigned long long hh = (unsigned long long)tab[z], pp = hh;
                int j = (int)(hh & (unsigned)nmask);
                for (;;) {
                    if (nt[j] < 0) { nt[j] = tab[z]; break; }
                    bool placed = false;
                    if (j + 9 <= nmask)
                        for (int q = 1; q <= 9; ++q)
                            if (nt[j + q] < 0) { nt[j + q] = tab[z]; placed = true; break; }
                    if (placed) break;
                    pp >>= 5;
                    j = (int)(((unsigned long long)j * 5ull + 1ull + pp) & (unsigned)nmask);
                }
            }
            tab = nt;
            mask = nmask;
        }
    }
    int c = 0;
    for (int z = 0; z <= mask; ++z)
        if (tab[z] >= 0) out[c++] = tab[z];
    return c;
}

#define TRI_WORDS 352  // >= ceil(10243/32), multiple of 32

// _trigram_candidates (quran_db.py:173-186) + the candidate-set rule of match_verse (:278-288),
// one block per utterance.  The transcript's distinct trigram ids are collected in a bitmap and
// compacted in ascending order; then the INVERTED index is walked: for every query trigram, in
// ascending id (= the canonical summation order), its IDF is added to the fp64 score of every
// verse of its posting list.  The per-verse scores live in LDS; wave w owns quarter w of the verse
// range (posting lists are verse-sorted and carry the four slice starts), so no two waves ever
// touch the same score and the loop needs no block barrier.  Work is sum(df) over the query's
// trigrams instead of a scan of all 420k forward postings per utterance.  Then: fewer than 20
// touched verses -> full scan; else the 50 highest sums, iterated in CPython set order.
__global__ __launch_bounds__(256) void k_trigram(QvTables tab, QvWork wk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *score = (double *)smem;                            // [N]
    double *sh_s = (double *)(score + tab.n_verses);           // [8]
    unsigned long long *sh_k = (unsigned long long *)(sh_s + 8);  // [8]
    int32_t *top = (int32_t *)(sh_k + 8);                      // [64]
    int32_t *tabA = top + 64;                                  // [256]
    int32_t *tabB = tabA + 256;                                // [256]
    uint32_t *tri_scratch = (uint32_t *)(tabB + 256);          // [272]
    double *sel_s = (double *)(tri_scratch + 272);             // [64]
    double *ord_s = sel_s + 64;                                // [64]
    int32_t *sel_p = (int32_t *)(ord_s + 64);                  // [64]
    uint32_t *bits = (uint32_t *)(sel_p + 64);                 // [TRI_WORDS]
    uint16_t *ids = (uint16_t *)(bits + TRI_WORDS);            // [QV_MAXQ]
    int32_t *cnt = (int32_t *)(ids + QV_MAXQ);                 // [8]: [0] number of ids, [1..4] touched per wave
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    QvUtt &u = wk.utt[b];
    const int m = u.q_len, N = tab.n_verses;
    if (m == 0) return;
    int32_t *cand1 = wk.cand1 + (size_t)b * N;
    if (u.force_full) {  // match_verse(use_trigram_index=False): every verse, in verse order
        for (int v = tid; v < N; v += 256) cand1[v] = v;
        if (tid == 0) { u.n_cand1 = N; u.full_scan = 1; }
        return;
    }
    for (int v = tid; v < N; v += 256) score[v] = -1.0;        // -1 = not touched
    for (int i = tid; i < TRI_WORDS; i += 256) bits[i] = 0;
    __syncthreads();
    const uint8_t *q = wk.q + (size_t)b * QV_MAXQ;
    for (int i = tid; i + 2 < m; i += 256) {
        // codes are < 64 (63 = outside the alphabet, never part of an indexed trigram)
        uint32_t key = ((uint32_t)q[i] << 12) | ((uint32_t)q[i + 1] << 6) | q[i + 2];
        int id = tab.tri_map[key];
        if (id == 0xFFFF) id = -1;
        if (id >= 0) atomicOr(&bits[id >> 5], 1u << (id & 31));
    }
    __syncthreads();
    if (wave == 0) {                                            // ascending list of the distinct ids
        int running = 0;
        for (int base = 0; base < TRI_WORDS; base += 64) {
            uint32_t w = base + lane < TRI_WORDS ? bits[base + lane] : 0u;
            int c = __popc(w), incl = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { int x = __shfl_up(incl, o); if (lane >= o) incl += x; }
            int pos = running + incl - c;
            while (w) { int bit = __ffs(w) - 1; w &= w - 1; ids[pos++] = (uint16_t)((base + lane) * 32 + bit); }
            running += __shfl(incl, 63);
        }
        if (lane == 0) cnt[0] = running;
    }
    __syncthreads();
    const int n_ids = cnt[0];
    for (int k0 = 0; k0 < n_ids; k0 += 64) {
        // this chunk's list bounds and IDFs: one memory round trip for 64 trigrams
        uint32_t P0 = 0, P1 = 0;
        double IDF = 0.0;
        if (k0 + lane < n_ids) {
            int id = ids[k0 + lane];
            P0 = tab.tri_slice[(size_t)id * 5 + wave];
            P1 = tab.tri_slice[(size_t)id * 5 + wave + 1];
            IDF = tab.tri_idf[id];
        }
        const int nchunk = n_ids - k0 < 64 ? n_ids - k0 : 64;
        for (int j0 = 0; j0 < nchunk; j0 += 8) {
            // the first 64 verses of 8 consecutive lists are requested together, then applied in list order
            int vv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                uint32_t p0 = __shfl(P0, (j0 + e) & 63), p1 = __shfl(P1, (j0 + e) & 63);
                vv[e] = (j0 + e < nchunk && p0 + lane < p1) ? (int)tab.tri_post[p0 + lane] : -1;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (j0 + e >= nchunk) break;
                const double idf = __shfl(IDF, (j0 + e) & 63);
                if (vv[e] >= 0) { double sv = score[vv[e]]; score[vv[e]] = sv < 0.0 ? idf : __dadd_rn(sv, idf); }
                uint32_t p0 = __shfl(P0, (j0 + e) & 63), p1 = __shfl(P1, (j0 + e) & 63);
                for (uint32_t base = p0 + 64; base < p1; base += 512) {   // long lists (frequent trigrams), 8 loads in flight
                    int lv[8];
#pragma unroll
                    for (int x = 0; x < 8; ++x) {
                        uint32_t p = base + x * 64 + lane;
                        lv[x] = p < p1 ? (int)tab.tri_post[p] : -1;
                    }
#pragma unroll
                    for (int x = 0; x < 8; ++x)
                        if (lv[x] >= 0) { double sv = score[lv[x]]; score[lv[x]] = sv < 0.0 ? idf : __dadd_rn(sv, idf); }
                }
            }
        }
    }
    __syncthreads();
    int local = 0;
    for (int v = tid; v < N; v += 256) local += score[v] >= 0.0;
    local = wave_sum_i(local);
    if (lane == 0) cnt[1 + wave] = local;
    __syncthreads();
    const int touched_total = cnt[1] + cnt[2] + cnt[3] + cnt[4];
    if (touched_total < 20) {  // quran_db.py:285-286
        for (int v = tid; v < N; v += 256) cand1[v] = v;
        if (tid == 0) { u.n_cand1 = N; u.full_scan = 1; }
        return;
    }
    int K = touched_total < 50 ? touched_total : 50;
    block_topk_select(score, N, K, tri_scratch, sel_p, sel_s, top, ord_s);
    if (tid == 0) {
        int n = pyset_order(top, K, cand1, tabA, tabB);
        u.n_cand1 = n;
        u.full_scan = 0;
    }
}

// ------------------------------------------------------------------ 4. fragment score --
// quran_db.py:211-237 (_fragment_score) with partial_ratio (:10-28) inlined.  One wave per
// (utterance, verse text).  mode 0: texts of the pass-1 iteration list; mode 1: every text
// of every gate-failed utterance (search()).
struct TextRef { const uint8_t *p; int n; int nw; int tid; };

__device__ __forceinline__ TextRef text_of(const QvTables &tab, int v, int variant) {
    TextRef t;
    if (variant == 0) { t.p = tab.clean + tab.clean_off[v]; t.n = tab.clean_len[v]; t.nw = tab.nw[0][v]; t.tid = v; }
    else if (variant == 1) { t.p = tab.alt + tab.alt_off[v]; t.n = tab.alt_len[v]; t.nw = tab.nw[1][v]; t.tid = tab.n_verses + v; }
    else {
        int nl = tab.nobsm_len[v];
        t.p = tab.clean + tab.clean_off[v] + tab.clean_len[v] - nl; t.n = nl; t.nw = tab.nw[2][v];
        t.tid = 2 * tab.n_verses + tab.nobsm_rank[v];
    }
    return t;
}

#define FRAG_DIRECT_MAX 128   // up to two rounds of windows: evaluate them all
#ifndef FRAG_STEP
#define FRAG_STEP 4          // anchor spacing of the coarse pass (power of two, >= 4)
#endif
#define FRAG_SCRATCH 2112      // int16 per wave: anchors [QV_MAXQ / 4 + 2] + refine list [QV_MAXQ]
#define FRAG_HEAVY 1200       // word-steps per lane above which a window scan counts as expensive (k_lcs_full's two-ended list)
#define FRAG_GRID 1024         // blocks of k_frag (4 waves each): four per CU, the list is consumed by whoever is free
#define FRAG_PM_STRIDE (QV_MAXW + 1)   // u64 per symbol row of the wave's LDS copy of the pattern masks (odd: rows spread over the banks)
__device__ __forceinline__ void frag_job(const QvTables &tab, const QvWork &wk, int b, int v, int variant, int lane, int16_t *scratch, uint64_t *lpm) {
    const QvUtt &u = wk.utt[b];
    double *out = wk.fs + ((size_t)b * tab.n_verses + v) * 3 + variant;
    if (variant == 2 && tab.nobsm_len[v] == 0) { if (lane == 0) *out = -1.0; return; }
    TextRef t = text_of(tab, v, variant);
    int m = u.q_len, n = t.n, qw = u.q_words, vw = t.nw;
    const uint8_t *q = wk.q + (size_t)b * QV_MAXQ;
    // " text " in " verse "  (word-boundary substring)
    bool sub = false;
    if (qw >= 3 && m <= n && wk.lcsf[((size_t)b * tab.n_verses + v) * 3 + variant] == m) {   // (needs LCS == m, see k_lcs_full)
        for (int i = lane; i + m <= n && !sub; i += 64) {
            if (i > 0 && t.p[i - 1] != 0) continue;
            if (i + m < n && t.p[i + m] != 0) continue;
            bool ok = true;
            for (int k = 0; k < m; ++k)
                if (q[k] != t.p[i + k] || q[k] >= QV_NSYM) { ok = false; break; }
            sub = ok;
        }
        sub = __any(sub);
    }
    bool windows = !sub && qw >= 4 && vw >= 2;
    // pattern = shorter string, text = (windows of) the longer one
    const uint64_t *pm;
    int stride, s, L;
    const uint8_t *lt;
    if (m <= n) { pm = wk.pm + (size_t)b * 2 * QV_NSYM * QV_MAXW; stride = QV_MAXW; s = m; lt = t.p; L = n; }
    else { pm = tab.pmv + tab.pmv_off[t.tid]; stride = qv_tmpl_w((n + 63) >> 6); s = n; lt = q; L = m; }
    int W = (s + 63) >> 6;
    int nwin = windows ? (L - s + 1) : 0;
    // full-string LCS comes from k_lcs_full (one text per lane); lanes here = sliding windows
    int full = wk.lcsf[((size_t)b * tab.n_verses + v) * 3 + variant], best = 0;
    // Only the maximum over the windows matters, and sliding the window by one position drops one
    // character and appends one, so LCS(w + d) <= LCS(w) + d.  With many windows, pass 0 evaluates
    // every 4th window (and the last one) and takes their maximum B; pass 1 evaluates exactly only
    // the windows whose bound from BOTH neighbouring anchors still exceeds B.  The result is the
    // exact maximum.  Up to FRAG_DIRECT_MAX windows are simply all evaluated in pass 0.
    // The window maximum only enters the score through blend(best), which is monotone in best and
    // never below the full-string ratio fr; and no window can beat the full string: best <= full.
    // If even blend(min(full, s)) == fr the windows cannot change the result and are skipped (the
    // usual case for a verse shorter than the transcript, whose word-count penalty outweighs any
    // window gain).  Otherwise the scan starts from the largest value that still blends to fr.
    const double fr = ratio_from(full, m, n);
    double pen = __ddiv_rn((double)vw, (double)(qw > 1 ? qw : 1));
    if (pen > 1.0) pen = 1.0;
    auto blend = [&](int x) -> double {
        double frag = ratio_from(x, s, s);
        if (!(frag > fr)) return fr;
        double blended = __dadd_rn(__dmul_rn(0.25, fr), __dmul_rn(__dmul_rn(0.75, frag), pen));
        return fr > blended ? fr : blended;
    };
    if (nwin > 0) {
        int ub = full < s ? full : s;
        if (blend(ub) == fr) nwin = 0;
        else if (nwin > FRAG_DIRECT_MAX) {
            int x = (int)(__ddiv_rn(fr, pen) * (double)s);         // estimate, then make it exact
            x = x < 0 ? 0 : (x > ub - 1 ? ub - 1 : x);
            while (x > 0 && blend(x) != fr) --x;
            while (x + 1 < ub && blend(x + 1) == fr) ++x;
            best = x;
        }
    }
    int16_t *cv = scratch, *list = scratch + FRAG_SCRATCH / 2;
    if (nwin > 0) {
        // Round 6: the pattern's match masks (40 symbols x W words) move into the wave's own LDS slice for the scan (row
        // QV_NSYM of the slice stays all zero: lcs_chunk<W, true>).  Every step of every window reads W mask words of ITS code:
        // ds_read instead of flat gathers over up to 40 global rows, and no select per mask word; the copy costs ~5 loads per
        // lane per item.
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < QV_NSYM * W; i += 64) {
            const int c = i / W, w = i - c * W;
            lpm[c * FRAG_PM_STRIDE + w] = pm[(size_t)c * stride + w];
        }
        QV_WSYNC();
    }
    const bool direct = nwin <= FRAG_DIRECT_MAX;
    const int nco = (nwin + FRAG_STEP - 1) / FRAG_STEP;       // anchors k * FRAG_STEP, k < nco; cv[nco] = last window
    int nlist = direct ? nwin : nco + 1;
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = lane; i < nlist; i += 64) {
            int w = pass ? (int)list[i] : (direct ? i : (i < nco ? FRAG_STEP * i : nwin - 1));
            int r = lcs_dispatch<true>(W, lpm, FRAG_PM_STRIDE, lt + w, s, s);
            if (!pass && !direct) cv[i] = (int16_t)r;
            best = max(best, r);
        }
        best = wave_max_i(best);
        if (direct || pass) break;
        __builtin_amdgcn_wave_barrier();
        nlist = 0;
        for (int base = 0; base < nwin; base += 64) {
            int w = base + lane;
            bool need = false;
            if (w < nwin - 1 && (w & (FRAG_STEP - 1)) != 0) {
                int k0 = w / FRAG_STEP, a1 = FRAG_STEP * (k0 + 1);
                int x0 = cv[k0], x1;
                if (a1 <= nwin - 1) x1 = cv[k0 + 1];
                else { a1 = nwin - 1; x1 = cv[nco]; }
                int ub = min(x0 + (w - FRAG_STEP * k0), x1 + (a1 - w));
                need = ub > best;
            }
            unsigned long long mask = __ballot(need);
            if (need) list[nlist + __popcll(mask & ((1ull << lane) - 1ull))] = (int16_t)w;
            nlist += __popcll(mask);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
        double res = fr;
        if (sub) res = fr > 0.98 ? fr : 0.98;
        else if (windows) res = blend(best);
        *out = res;
    }
}

// full-string LCS(transcript, text), one text per lane, pattern = the transcript's match masks, AND everything
// _fragment_score decides before it looks at a single window (quran_db.py:211-237): the word-boundary substring test,
// "fewer than 4 query words / fewer than 2 verse words", and the exact bound "no window can lift the blend above the
// full-string ratio" (frag_job's comment).  Nine texts in ten are final here, one text per LANE; only the rest go on the
// fragment work list that k_frag's waves consume (one text per WAVE, lanes = windows).  The first version ran one wave
// per text for all of them: 64 lanes repeating the same scalar decision for 12,472 texts per gate-failed utterance.
// mode 0: the texts of the pass-1 iteration list; mode 1 (slow-list utterances): clean + alt of every verse for search()
// (skipped when pass 1 already scanned everything) and, as a third job per verse, pass 3's LCS of the spaceless
// transcript against the clean text (c2c-direct/run.py:284-297; a space matches nothing in the spaceless pattern, so
// the spaced text is streamed).
__global__ __launch_bounds__(256) void k_lcs_full(QvTables tab, QvWork wk, int mode) {
    // grid = (utterance, chunk of the job list): the hardware dispatches workgroups x-fastest, and in mode 1 chunk 0 holds
    // every utterance's LONGEST verses (len_order).  With the chunk on x the 64 long blocks were spread over the whole
    // launch and the last of them started when everything else was done; on y they all start first (longest first).
    const int slot = blockIdx.x, chunk = blockIdx.y, nchunk = gridDim.y;
    int b;
    if (mode == 0) b = slot;
    else { if (slot >= *wk.n_fail) return; b = wk.fail_list[slot]; }
    const QvUtt &u = wk.utt[b];
    if (u.q_len == 0) return;
    const int m = u.q_len, ms = u.qs_len, W = (m + 63) >> 6, N = tab.n_verses, qw = u.q_words;
    __shared__ uint64_t spm[2 * QV_PM_ROWS * QV_PMS];
    load_pm_lds(spm, wk.pm + (size_t)b * 2 * QV_NSYM * QV_MAXW);
    const uint64_t *pm = spm;
    const int32_t *cand1 = wk.cand1 + (size_t)b * N;
    int16_t *out = wk.lcsf + (size_t)b * N * 3;
    const bool short_q = qw < 4;
    const uint8_t *q = wk.q + (size_t)b * QV_MAXQ;
    const int lane = threadIdx.x & 63;
    const int jobs = mode == 0 ? u.n_cand1 * 3 : N * 3;
    for (int j = chunk * 256 + threadIdx.x; j < jobs; j += nchunk * 256) {
        int v, variant;
        if (mode == 0) { v = cand1[j / 3]; variant = j % 3; }
        else {
            v = tab.len_order[j / 3];   // similar lengths per wave
            variant = j % 3;
            if (variant == 2) {         // pass 3
                wk.lcs_p3[(size_t)b * N + v] =
                    (int16_t)lcs_dispatch<true>((ms + 63) >> 6, spm + QV_PM_ROWS * QV_PMS, QV_PMS, tab.clean + tab.clean_off[v], tab.clean_len[v], ms);
                continue;
            }
            if (u.full_scan) continue;  // search(): pass 1 already scored every verse
        }
        double *fs = wk.fs + ((size_t)b * N + v) * 3 + variant;
        if (variant == 2 && tab.nobsm_len[v] == 0) { *fs = -1.0; continue; }
        TextRef t = text_of(tab, v, variant);
        const int l = lcs_dispatch<true>(W, pm, QV_PMS, t.p, t.n, m);
        out[v * 3 + variant] = (int16_t)l;
        const int n = t.n, vw = t.nw;
        const double fr = ratio_from(l, m, n);
        bool sub = false;   // " text " in " verse "  (word-boundary substring)
        // (a substring is a common subsequence of full length: only texts with LCS == m are scanned at all)
        if (qw >= 3 && m <= n && l == m) {
            for (int i = 0; i + m <= n && !sub; ++i) {
                if (i > 0 && t.p[i - 1] != 0) continue;
                if (i + m < n && t.p[i + m] != 0) continue;
                bool ok = true;
                for (int k = 0; k < m; ++k)
                    if (q[k] != t.p[i + k] || q[k] >= QV_NSYM) { ok = false; break; }
                sub = ok;
            }
        }
        bool need = false, heavy = false;
        if (sub) *fs = fr > 0.98 ? fr : 0.98;
        else if (short_q || vw < 2) *fs = fr;                 // no windows below 4 query words / 2 verse words
        else {
            const int s = m <= n ? m : n;
            double pen = __ddiv_rn((double)vw, (double)(qw > 1 ? qw : 1));
            if (pen > 1.0) pen = 1.0;
            const int ub = l < s ? l : s;                      // no window beats the full string
            const double frag = ratio_from(ub, s, s);
            double bl = fr;
            if (frag > fr) {
                const double blended = __dadd_rn(__dmul_rn(0.25, fr), __dmul_rn(__dmul_rn(0.75, frag), pen));
                bl = fr > blended ? fr : blended;
            }
            if (bl == fr) *fs = fr;
            else {
                need = true;
                // Round 6: what the scan will cost (rounds of 64 windows x pattern length x words), so that k_frag can start the
                // expensive items first: they go to the FRONT of the list, the cheap ones fill it from the BACK, and the queue is
                // consumed front to back -- the longest scans no longer start when everything else is done
                const int nwin = (m <= n ? n : m) - s + 1, nev = nwin <= FRAG_DIRECT_MAX ? nwin : (nwin + FRAG_STEP - 1) / FRAG_STEP + 1;
                heavy = ((nev + 63) >> 6) * s * ((s + 63) >> 6) >= FRAG_HEAVY;
            }
        }
        // wave-aggregated append to the work list (two ends)
        const unsigned long long mh = __ballot(need && heavy), ml = __ballot(need && !heavy);
        if (need) {
            const unsigned long long mask = heavy ? mh : ml;
            const int leader = __ffsll((long long)mask) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&wk.frag_ctr[heavy ? 0 : 2], __popcll(mask));
            base = __shfl(base, leader);
            const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
            wk.frag_list[heavy ? pos : wk.frag_cap - 1 - pos] = ((uint32_t)b << 15) | ((uint32_t)v << 2) | (uint32_t)variant;
        }
    }
}

// the window scans: one text per wave, taken off the work list k_lcs_full left.  A wave's first item is its own index;
// further ones come from a shared cursor, so the waves of a launch that finds few items leave without an atomic and a
// long text (hundreds of windows of a 16-word pattern) does not hold a fixed share of the list hostage.
__global__ __launch_bounds__(256) void k_frag(QvTables tab, QvWork wk) {
    __shared__ int16_t frag_scratch[4][FRAG_SCRATCH];
    __shared__ uint64_t frag_pm[4][QV_PM_ROWS * FRAG_PM_STRIDE];
    int16_t *scratch = frag_scratch[threadIdx.x >> 6];
    uint64_t *lpm = frag_pm[threadIdx.x >> 6];
    for (int i = threadIdx.x & 63; i < FRAG_PM_STRIDE; i += 64) lpm[QV_NSYM * FRAG_PM_STRIDE + i] = 0ull;   // the all-zero row
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwave = gridDim.x * 4;
    const int n_heavy = wk.frag_ctr[0], n_items = n_heavy + wk.frag_ctr[2];
    int idx = wave;
    while (idx < n_items) {
        const uint32_t it = wk.frag_list[idx < n_heavy ? idx : wk.frag_cap - 1 - (idx - n_heavy)];
        frag_job(tab, wk, (int)(it >> 15), (int)((it >> 2) & 0x1FFFu), (int)(it & 3u), lane, scratch, lpm);
        int nx = 0;
        if (lane == 0) nx = atomicAdd(&wk.frag_ctr[1], 1);
        idx = nwave + __shfl(nx, 0);
    }
}

// ------------------------------------------------------------------ 4b. hint scores ----
// _suffix_prefix_score (quran_db.py:188-208) for the <= 3 verses of the continuation hint, both
// of their texts: the transcript minus its first 1..min(words/2, 4) words against the equally
// long word prefix of the verse.  One lane per (verse, text, trim); pattern = the verse's
// precomputed match masks (a word prefix of the pattern is the low bits of the same masks).
__global__ __launch_bounds__(64) void k_hint_sp(QvTables tab, QvWork wk, int b) {
    QvUtt &u = wk.utt[b];
    const int lane = threadIdx.x, m = u.q_len, qw = u.q_words;
    const uint8_t *q = wk.q + (size_t)b * QV_MAXQ;
    double best = 0.0;
    const int k = lane >> 3, variant = (lane >> 2) & 1, trim = (lane & 3) + 1;   // 3 x 2 x 4 jobs
    const int max_trim = (qw >> 1) < 4 ? (qw >> 1) : 4;
    if (k < u.hint_n && m > 0 && qw >= 2 && trim <= max_trim) {
        const int v = u.hint_v[k];
        TextRef t = text_of(tab, v, variant);
        if (t.nw >= 2) {
            int start = 0, seen = 0;                         // transcript after its first `trim` words
            for (int i = 0; i < m; ++i)
                if (q[i] == 0 && ++seen == trim) { start = i + 1; break; }
            const int nt = qw - trim, pw = nt < t.nw ? nt : t.nw;
            int plen = t.n;
            if (pw < t.nw) {
                seen = 0;
                for (int i = 0; i < t.n; ++i)
                    if (t.p[i] == 0 && ++seen == pw) { plen = i; break; }
            }
            const int ls = m - start;
            const uint64_t *pm = tab.pmv + tab.pmv_off[t.tid];
            int l = lcs_dispatch((plen + 63) >> 6, pm, qv_tmpl_w((t.n + 63) >> 6), q + start, ls, plen);
            best = ratio_from(l, ls, plen);
        }
    }
    // max over the 8 lanes of a verse
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { double x = __shfl_xor(best, o); if (x > best) best = x; }
    if ((lane & 7) == 0 && k < u.hint_n) u.hint_sp[k] = best;
}

__global__ void k_set_hint(QvWork wk, int b, int n, int v0, int v1, int v2, double b0, double b1, double b2) {
    QvUtt &u = wk.utt[b];
    u.force_full = 1;
    u.hint_n = n;
    u.hint_v[0] = v0; u.hint_v[1] = v1; u.hint_v[2] = v2;
    u.hint_bonus[0] = b0; u.hint_bonus[1] = b1; u.hint_bonus[2] = b2;
    u.hint_sp[0] = u.hint_sp[1] = u.hint_sp[2] = 0.0;
}

// ------------------------------------------------------------------ 5. pass-1 finalize -
// stable descending sort of min(raw,1.0) in iteration order (quran_db.py:290-331): only the
// first max(top_k,5) entries are ever used, selected by repeated block argmax.
__global__ __launch_bounds__(256) void k_pass1_final(QvTables tab, QvWork wk, QvKnobs kn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *sc = (double *)smem;  // [n_cand1]
    double *sh_s = sc + tab.n_verses;
    unsigned long long *sh_k = (unsigned long long *)(sh_s + 8);
    int b = blockIdx.x, tid = threadIdx.x;
    QvUtt &u = wk.utt[b];
    if (u.q_len == 0) return;
    int n = u.n_cand1, N = tab.n_verses;
    const int32_t *cand1 = wk.cand1 + (size_t)b * N;
    const double *fs = wk.fs + (size_t)b * N * 3;
    for (int p = tid; p < n; p += 256) {
        int v = cand1[p];
        double a = fs[v * 3], c = fs[v * 3 + 1], d = fs[v * 3 + 2];
        double raw = a > c ? a : c;
        if (d > raw) raw = d;
        double bonus = 0.0;                       // continuation hint (quran_db.py:300-311); none on the hot path
        for (int k = 0; k < u.hint_n; ++k)
            if (v == u.hint_v[k]) { if (u.hint_sp[k] > raw) raw = u.hint_sp[k]; bonus = u.hint_bonus[k]; }
        double tot = __dadd_rn(raw, bonus);
        sc[p] = tot < 1.0 ? tot : 1.0;
    }
    __syncthreads();
    int K = kn.top_text > 5 ? kn.top_text : 5;
    if (K > n) K = n;
    if (K > QV_RUNNER_CAP) K = QV_RUNNER_CAP;
    int K20 = n < 20 ? n : 20;
    int rounds = K > K20 ? K : K20;
    int32_t *ridx = wk.runner_idx + (size_t)b * QV_RUNNER_CAP;
    double *rsc = wk.runner_score + (size_t)b * QV_RUNNER_CAP;
    double *sel_s = sh_s;                                 // [128]
    int32_t *sel_p = (int32_t *)(sel_s + QV_RUNNER_CAP);  // [128]
    uint32_t *scratch = (uint32_t *)(sel_p + QV_RUNNER_CAP);  // [272]
    int32_t *ord_p = (int32_t *)(scratch + 272);          // [128]
    double *ord_s = (double *)(ord_p + QV_RUNNER_CAP);    // [128]
    int nsel = block_topk_select(sc, n, rounds, scratch, sel_p, sel_s, ord_p, ord_s);
    for (int r = tid; r < nsel; r += 256) { ridx[r] = cand1[ord_p[r]]; rsc[r] = ord_s[r]; }
    if (tid == 0) {
        u.best1_idx = cand1[ord_p[0]];
        u.best1_score = ord_s[0];
        int ns = 0;
        for (int r = 0; r < K20; ++r) {
            int su = tab.surah[cand1[ord_p[r]]];
            bool seen = false;
            for (int i = 0; i < ns; ++i) seen |= (u.surah20[i] == su);
            if (!seen) u.surah20[ns++] = su;
        }
        u.n_surah20 = ns;
    }
    if (tid == 0) u.n_runners = K < kn.top_text ? K : kn.top_text;
}


// ------------------------------------------------------------------ 6. span pass -------
// (Measured and rejected, round 3: one START verse per lane with all its spans read off ONE walk over the longest live
// span -- the span texts of a start verse are prefixes of one another and the recurrence walks front to back, lcs_feed /
// lcs_count -- does 3.3x fewer word-steps and reproduces every fixture, but ran 2.4x SLOWER than this kernel at batch 1
// and batch 64 alike (594 vs 235 us, 2,030 vs 840 us), with per-ayah loops, with one loop cut at the boundaries, and with
// the span ends held in registers.  Five times fewer lanes, each carrying the longest chain: the pass is bound by the
// serial add-with-carry chain per lane, and the compiler schedules the plain counted loop below far better.)
// quran_db.py:334-365: every window of 2..max_span ayat of the surahs of the top 20.  One
// span text per lane; texts are contiguous slices of the padded clean array.  A span is
// skipped when even LCS = min(m, n) could not beat the pass-1 best (exact pruning).
__device__ void base_final_one(const QvTables &tab, const QvWork &wk, const QvKnobs &kn, int b, int force_ctc);

__global__ __launch_bounds__(256) void k_spans(QvTables tab, QvWork wk, QvKnobs kn) {
    __shared__ double sh_s[8];
    __shared__ unsigned long long sh_k[8];
    // grid = (block of the utterance's job space, utterance).  (Round 4: the transposed grid that cut k_lcs_full's tail --
    // utterance on x, so that every round of blocks mixes all utterances -- made THIS kernel slower, 738 -> 848 us per
    // launch: its blocks are balanced inside an utterance already, and neighbours then no longer share an utterance's
    // tables in L2.)
    const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x, tid = threadIdx.x;
    // pass 1's window scans are done, search()'s have not started: empty the fragment work list
    if (blk == 0 && b == 0 && tid == 0) { wk.frag_ctr[0] = 0; wk.frag_ctr[1] = 0; wk.frag_ctr[2] = 0; }
    const QvUtt &u = wk.utt[b];
    double best = -1.0;
    unsigned long long bkey = ~0ull;
    __shared__ uint64_t spm[2 * QV_PM_ROWS * QV_PMS];
    load_pm_lds(spm, wk.pm + (size_t)b * 2 * QV_NSYM * QV_MAXW);
    if (u.q_len > 0) {
        int m = u.q_len, W = (m + 63) >> 6;
        const uint64_t *pm = spm;
        int per = kn.max_span - 1;
        // One flat job space over the surahs of the top 20, so that every lane of the grid
        // has a span (the surahs are not walked one after the other).  Inside a surah the
        // lanes are span-major - neighbouring lanes hold the same number of ayat starting at
        // consecutive verses, i.e. texts of similar length, which is what bounds a wave.  The
        // tie-break key stays the reference's iteration rank: surahs in order, then start
        // ayah, then span length.
        int cum[21];
        cum[0] = 0;
#pragma unroll
        for (int si = 0; si < 20; ++si)
            cum[si + 1] = cum[si] + (si < u.n_surah20 ? tab.surah_len[u.surah20[si] - 1] * per : 0);
        const int total = cum[20];
        // job g -> (span text, bound check); false = nothing to score (window runs off the surah / cannot beat pass 1)
        auto job = [&](int g, uint32_t &start, int &n, double &bonus, unsigned long long &key) -> bool {
            int si = 0, base = 0;
#pragma unroll
            for (int k = 1; k < 20; ++k)
                if (g >= cum[k]) { si = k; base = cum[k]; }
            int s = u.surah20[si];
            int s0 = tab.surah_start[s - 1], sl = tab.surah_len[s - 1];
            int r = g - base;
            int span = 2 + r / sl, i = r % sl;
            if (i + span > sl) return false;
            int v0 = s0 + i, v1 = v0 + span - 1;
            int nl = tab.nobsm_len[v0];
            start = tab.clean_off[v0] + (nl ? tab.clean_len[v0] - nl : 0);
            n = (int)(tab.clean_off[v1] + tab.clean_len[v1] - start);
            int mn = m < n ? m : n;
            bonus = 0.0;                          // of the span's first verse (quran_db.py:352-353)
            for (int k = 0; k < u.hint_n; ++k)
                if (v0 == u.hint_v[k]) bonus = u.hint_bonus[k];
            double ub = __dadd_rn(ratio_from(mn, m, n), bonus);
            if (!((ub < 1.0 ? ub : 1.0) > u.best1_score)) return false;
            key = (unsigned long long)base + (unsigned long long)(i * per + span - 2);
            return true;
        };
        auto score = [&](int l, int n, double bonus, unsigned long long key) {
            double raw = __dadd_rn(ratio_from(l, m, n), bonus);
            double sc = raw < 1.0 ? raw : 1.0;
            if (better(sc, key, best, bkey)) { best = sc; bkey = key; }
        };
        if (W <= 2) {
            // short transcripts: one span per lane (the chain is at most two words per code)
            for (int g = blk * 256 + tid; g < total; g += nblk * 256) {
                uint32_t start; int n; double bonus; unsigned long long key;
                if (!job(g, start, n, bonus, key)) continue;
                score(lcs_dispatch(W, pm, QV_PMS, tab.clean + start, n, m), n, bonus, key);
            }
        } else {
            // long transcripts: G lanes per span (lcs_systolic).  Every group first walks to its next span that survives
            // the bound -- most do not -- so that the groups of a wave enter the recurrence together.
            auto run = [&](auto g_c) {
                constexpr int G = decltype(g_c)::value;
                const int w = tid & (G - 1), ngroups = nblk * (256 / G);
                for (int g = blk * (256 / G) + tid / G; g < total; g += ngroups) {
                    uint32_t start = 0; int n = 0; double bonus = 0.0; unsigned long long key = 0;
                    bool ok = false;
                    for (; g < total; g += ngroups)
                        if ((ok = job(g, start, n, bonus, key))) break;
                    if (!ok) break;
                    const int l = lcs_systolic<G>(pm, QV_PMS, tab.clean + start, n, m, W, w);
                    if (w == 0) score(l, n, bonus, key);
                }
            };
            if (W <= 4) run(std::integral_constant<int, 4>{});
            else if (W <= 8) run(std::integral_constant<int, 8>{});
            else run(std::integral_constant<int, 16>{});
        }
    }
    block_best(best, bkey, sh_s, sh_k);
    if (tid == 0) {
        wk.span_part_score[(size_t)b * QV_SPAN_BLOCKS + blk] = best;
        wk.span_part_key[(size_t)b * QV_SPAN_BLOCKS + blk] = bkey;
    }
}

// ---- k_spans2 (round 5): the span pass with PREFIX SHARING.  The spans that start at verse v0 -- 2, 3, .. max_span ayat --
// are prefixes of one another, and the bit-parallel LCS state after n codes of a text IS the state of its n-code prefix:
// one walk over the longest surviving span of a start verse, with the LCS read off at every ayah end on the way, replaces
// (max_span - 1) walks of 2 + 3 + .. ayat (3.3 x fewer word-steps at max_span = 6).  The walk runs over tab.clean8, where
// every verse is padded to whole 8-code chunks with codes that match nothing, so that an ayah end is a chunk end; a span
// whose first verse loses its bismillah starts inside a chunk and masks the codes in front of it.  Same spans, same
// exact bound, same scores and tie-break keys as k_spans (qv_debug_kernel_variant(2, 0) selects the old kernel;
// tests/test_gpu_postlogits.py compares the two).
struct SpanJob {
    int v0, emax;              // start verse; longest span (in ayat) that survives the bound, 0 = none
    unsigned surv;             // bit e: the span of e ayat survives
    uint32_t start;            // its text in tab.clean
    uint32_t a0;               // first chunk in tab.clean8 (multiple of 8)
    int lead;                  // codes of that chunk in front of the text
    double bonus;
    unsigned long long key0;   // tie-break key of the span of 2 ayat (the span of e: key0 + e - 2)
};

// the pattern's words spread over G lanes as in lcs_systolic, but skewed by whole CHUNKS: at chunk-step t lane w runs the
// 8 codes of chunk t - w with the 8 carries lane w - 1 produced for that chunk one step earlier (one DPP per chunk instead
// of one per code; every lane is chunk-aligned, so "an ayah ends here" is one compare per chunk).  snap packs the lane's
// zero counts at the ends of ayat 2 .. (7 bits each); all G lanes pass the same job.
template <int G>
__device__ __forceinline__ uint64_t lcs_systolic_chunks(const uint64_t *__restrict__ pm, int stride, const uint8_t *__restrict__ c8,
                                                        const uint32_t *__restrict__ off8, const SpanJob &j, int m, int W, int w) {
    uint64_t V = ~0ull, snap = 0;
    unsigned cout8 = 0;
    const bool act = w < W;
    const int lo = w * 64;
    const uint64_t zmask = !act ? 0ull : (m >= lo + 64 ? ~0ull : (m > lo ? ((1ull << (m - lo)) - 1ull) : 0ull));
    const int nch = (int)(off8[j.v0 + j.emax] - j.a0) >> 3;
    int e = 1, next_end = (int)(off8[j.v0 + 1] - j.a0) >> 3;       // chunks up to the end of ayah e
    for (int t = 0; t < nch + G - 1; ++t) {
        const int c = t - w;
        unsigned cin8 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)cout8, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
        if (w == 0) cin8 = 0;
        uint64_t chunk = ~0ull;                 // filler: matches nothing
        if (c >= 0 && c < nch) chunk = *(const uint64_t *)(c8 + j.a0 + 8 * (size_t)c);
        if (c == 0 && j.lead) chunk |= (1ull << (8 * j.lead)) - 1ull;
        uint64_t mk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int code = (int)((chunk >> (8 * k)) & 0xFF);
            const bool valid = act && code < QV_NSYM;
            // (pm is always load_pm_lds's copy here: invalid codes and idle lanes read its all-zero row -- a zero mask leaves V
            // alone and produces no carry)
            mk[k] = pm[(size_t)(valid ? code : QV_NSYM) * stride + (act ? w : 0)];
        }
        cout8 = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsigned long long carry;
            const uint64_t s2 = __builtin_addcll(V, V & mk[k], (unsigned long long)((cin8 >> k) & 1u), &carry);
            V = s2 | (V & ~mk[k]);
            cout8 |= (unsigned)carry << k;
        }
        if (c + 1 == next_end) {                // the ayah ends with this chunk (rare: <= max_span times per walk)
            if (e >= 2) snap |= (uint64_t)__popcll(~V & zmask) << (7 * (e - 2));
            ++e;
            // (past the longest surviving span nothing is read off any more: the skewed lanes keep stepping for up to G - 1
            // chunks after the walk's last one, and further ayah ends there must not shift into the fields above)
            next_end = e <= j.emax ? (int)(off8[j.v0 + e] - j.a0) >> 3 : 0x7FFFFFFF;
        }
    }
    return snap;
}

__global__ __launch_bounds__(256) void k_spans2(QvTables tab, QvWork wk, QvKnobs kn) {
    __shared__ double sh_s[8];
    __shared__ unsigned long long sh_k[8];
    const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x, tid = threadIdx.x;
    // pass 1's window scans are done, search()'s have not started: empty the fragment work list
    if (blk == 0 && b == 0 && tid == 0) { wk.frag_ctr[0] = 0; wk.frag_ctr[1] = 0; wk.frag_ctr[2] = 0; }
    const QvUtt &u = wk.utt[b];
    double best = -1.0;
    unsigned long long bkey = ~0ull;
    __shared__ uint64_t spm[2 * QV_PM_ROWS * QV_PMS];
    load_pm_lds(spm, wk.pm + (size_t)b * 2 * QV_NSYM * QV_MAXW);
    if (u.q_len > 0) {
        const int m = u.q_len, W = (m + 63) >> 6;
        const uint64_t *pm = spm;
        const int per = kn.max_span - 1;
        // job space: one job per START verse of the surahs of the top 20; cum1 counts jobs, cumk the old kernel's
        // (start, span) ranks that the tie-break keys are made of
        int cum1[21];
        cum1[0] = 0;
#pragma unroll
        for (int si = 0; si < 20; ++si) cum1[si + 1] = cum1[si] + (si < u.n_surah20 ? tab.surah_len[u.surah20[si] - 1] : 0);
        const int total = cum1[20];
        auto job = [&](int g, SpanJob &j) -> bool {
            int si = 0, base = 0;
#pragma unroll
            for (int k = 1; k < 20; ++k)
                if (g >= cum1[k]) { si = k; base = cum1[k]; }
            const int s = u.surah20[si];
            const int s0 = tab.surah_start[s - 1], sl = tab.surah_len[s - 1];
            const int i = g - base;
            j.v0 = s0 + i;
            const int nl = tab.nobsm_len[j.v0];
            const uint32_t skip = nl ? tab.clean_len[j.v0] - nl : 0;
            j.start = tab.clean_off[j.v0] + skip;
            j.bonus = 0.0;                        // of the span's first verse (quran_db.py:352-353)
            for (int k = 0; k < u.hint_n; ++k)
                if (j.v0 == u.hint_v[k]) j.bonus = u.hint_bonus[k];
            j.surv = 0; j.emax = 0;
            for (int e = 2; e <= kn.max_span && i + e <= sl; ++e) {
                const int v1 = j.v0 + e - 1;
                const int n = (int)(tab.clean_off[v1] + tab.clean_len[v1] - j.start);
                const int mn = m < n ? m : n;
                const double ub = __dadd_rn(ratio_from(mn, m, n), j.bonus);
                if ((ub < 1.0 ? ub : 1.0) > u.best1_score) { j.surv |= 1u << e; j.emax = e; }
            }
            if (!j.emax) return false;
            const uint32_t s8 = tab.clean8_off[j.v0] + 1 + skip;
            j.a0 = s8 & ~7u;
            j.lead = (int)(s8 - j.a0);
            j.key0 = (unsigned long long)(base * per) + (unsigned long long)(i * per);
            return true;
        };
        auto score = [&](int l, const SpanJob &j, int e) {
            const int v1 = j.v0 + e - 1;
            const int n = (int)(tab.clean_off[v1] + tab.clean_len[v1] - j.start);
            const double raw = __dadd_rn(ratio_from(l, m, n), j.bonus);
            const double sc = raw < 1.0 ? raw : 1.0;
            const unsigned long long key = j.key0 + (unsigned long long)(e - 2);
            if (better(sc, key, best, bkey)) { best = sc; bkey = key; }
        };
        if (W <= 2) {
            // short transcripts: one start verse per lane
            auto walk = [&](auto w_c) {
                constexpr int WW = decltype(w_c)::value;
                for (int g = blk * 256 + tid; g < total; g += nblk * 256) {
                    SpanJob j;
                    if (!job(g, j)) continue;
                    uint64_t V[WW];
#pragma unroll
                    for (int k = 0; k < WW; ++k) V[k] = ~0ull;
                    uint32_t pos = j.a0;
                    uint64_t next = *(const uint64_t *)(tab.clean8 + pos);
                    if (j.lead) next |= (1ull << (8 * j.lead)) - 1ull;
                    for (int e = 1; e <= j.emax; ++e) {
                        const uint32_t end = tab.clean8_off[j.v0 + e];
                        for (; pos < end; pos += 8) {
                            const uint64_t chunk = next;
                            next = *(const uint64_t *)(tab.clean8 + pos + 8);       // (the array is padded by 64 filler codes)
                            lcs_chunk<WW, true>(V, pm, QV_PMS, chunk, 8);
                        }
                        if (j.surv >> e & 1u) score(lcs_count<WW>(V, m), j, e);
                    }
                }
            };
            if (W <= 1) walk(std::integral_constant<int, 1>{});
            else walk(std::integral_constant<int, 2>{});
        } else {
            // long transcripts: G lanes per start verse; every group first walks to its next job that survives the bound,
            // so that the groups of a wave enter the recurrence together
            auto run = [&](auto g_c) {
                constexpr int G = decltype(g_c)::value;
                const int w = tid & (G - 1), ngroups = nblk * (256 / G);
                for (int g = blk * (256 / G) + tid / G; g < total; g += ngroups) {
                    SpanJob j;
                    bool ok = false;
                    for (; g < total; g += ngroups)
                        if ((ok = job(g, j))) break;
                    if (!ok) break;
                    uint64_t snap = lcs_systolic_chunks<G>(pm, QV_PMS, tab.clean8, tab.clean8_off, j, m, W, w);
                    for (int e = 2; e <= j.emax; ++e) {
                        int cnt = (int)((snap >> (7 * (e - 2))) & 0x7F);
#pragma unroll
                        for (int o = G / 2; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
                        if (w == 0 && (j.surv >> e & 1u)) score(cnt, j, e);
                    }
                }
            };
            if (W <= 4) run(std::integral_constant<int, 4>{});
            else if (W <= 8) run(std::integral_constant<int, 8>{});
            else run(std::integral_constant<int, 16>{});
        }
    }
    block_best(best, bkey, sh_s, sh_k);
    if (tid == 0) {
        wk.span_part_score[(size_t)b * QV_SPAN_BLOCKS + blk] = best;
        wk.span_part_key[(size_t)b * QV_SPAN_BLOCKS + blk] = bkey;
    }
}

__device__ void base_final_one(const QvTables &tab, const QvWork &wk, const QvKnobs &kn, int b, int force_ctc);

// (Measured and rejected: folding this -- and the candidate assembly, and the final decision -- into the tail of the
// kernel in front of it with a "last block done" counter.  Publishing a block's results to a block on another XCD takes an
// agent-scope release, i.e. an L2 write-back per block on this 8-XCD part: k_spans 23 -> 105 us, k_topk 57 -> 108 us,
// k_ctc 143 -> 601 us.  A kernel boundary is the cheap device-wide synchronisation here.)
__global__ void k_base_final(QvTables tab, QvWork wk, QvKnobs kn, int batch, int force_ctc) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) base_final_one(tab, wk, kn, b, force_ctc);
}

// base = better of pass-1 best and span best; gate; result for the text branch.
__device__ void base_final_one(const QvTables &tab, const QvWork &wk, const QvKnobs &kn, int b, int force_ctc) {
    QvUtt &u = wk.utt[b];
    if (u.q_len == 0) return;
    double best = -1.0;
    unsigned long long key = ~0ull;
    for (int i = 0; i < QV_SPAN_BLOCKS; ++i) {
        double s = wk.span_part_score[(size_t)b * QV_SPAN_BLOCKS + i];
        unsigned long long k = wk.span_part_key[(size_t)b * QV_SPAN_BLOCKS + i];
        if (better(s, k, best, key)) { best = s; key = k; }
    }
    u.base_start = u.best1_idx; u.base_span = 1; u.base_score = u.best1_score;
    if (best > u.best1_score) {
        int per = kn.max_span - 1;
        unsigned long long j = key;
        for (int si = 0; si < u.n_surah20; ++si) {
            int s = u.surah20[si];
            unsigned long long jobs = (unsigned long long)tab.surah_len[s - 1] * per;
            if (j < jobs) {
                u.base_start = tab.surah_start[s - 1] + (int)(j / per);
                u.base_span = 2 + (int)(j % per);
                u.base_score = best;
                break;
            }
            j -= jobs;
        }
    }
    u.use_ctc = (u.base_score < kn.threshold) || force_ctc;
    if (u.use_ctc) u.flags |= QV_FLAG_USED_CTC;
    // the "slow list": utterances whose search()/pass-3/candidate stages must run.  With
    // skip_unused == 0 that is every utterance, as in the reference (mixed/run.py:84 precedes
    // the gate at :96); their output is only consumed when use_ctc is set.
    if (u.use_ctc || !kn.skip_unused) {
        int slot = atomicAdd(wk.n_fail, 1);
        wk.fail_list[slot] = b;
    }
}

// ------------------------------------------------------------------ 7. pass 3 ----------
// c2c-direct/run.py:284-297: max(ratio(t, clean), ratio(t.spaceless, clean.spaceless)).  Both LCS values come from
// k_lcs_full (lcsf / lcs_p3); the two ratios and their maximum are formed where they are consumed, in k_topk.
// top-k (score desc, verse index asc) of search() and pass 3
__global__ __launch_bounds__(256) void k_topk(QvTables tab, QvWork wk, QvKnobs kn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *sc = (double *)smem;
    double *sh_s = sc + tab.n_verses;
    unsigned long long *sh_k = (unsigned long long *)(sh_s + 8);
    if ((int)blockIdx.x >= *wk.n_fail) return;
    int b = wk.fail_list[blockIdx.x], which = blockIdx.y, tid = threadIdx.x;
    int N = tab.n_verses;
    if (wk.utt[b].q_len == 0) return;
    if (which == 0) {
        const double *fs = wk.fs + (size_t)b * N * 3;
        for (int v = tid; v < N; v += 256) { double a = fs[v * 3], c = fs[v * 3 + 1]; sc[v] = a > c ? a : c; }
    } else {
        const QvUtt &u = wk.utt[b];
        const int m = u.q_len, ms = u.qs_len;
        for (int v = tid; v < N; v += 256) {
            const int n = tab.clean_len[v], ns = n - (tab.nw[0][v] - 1);
            const double a = ratio_from(wk.lcsf[((size_t)b * N + v) * 3], m, n), c = ratio_from(wk.lcs_p3[(size_t)b * N + v], ms, ns);
            sc[v] = a > c ? a : c;
        }
    }
    __syncthreads();
    int K = kn.top_text < QV_RUNNER_CAP ? kn.top_text : QV_RUNNER_CAP;
    if (K > N) K = N;
    int32_t *oi = (which == 0 ? wk.top_search : wk.top_p3) + (size_t)b * QV_RUNNER_CAP;
    double *os = (which == 0 ? wk.top_search_sc : wk.top_p3_sc) + (size_t)b * QV_RUNNER_CAP;
    double *sel_s = sh_s;                                 // [128]
    int32_t *sel_p = (int32_t *)(sel_s + QV_RUNNER_CAP);  // [128]
    uint32_t *scratch = (uint32_t *)(sel_p + QV_RUNNER_CAP);
    block_topk_select(sc, N, K, scratch, sel_p, sel_s, oi, os);
}


// ------------------------------------------------------------------ 8. candidates ------
// c2c-direct/run.py:251-311: ordered, de-duplicated union + span expansion of the first
// top_span_refs single refs.  The order defines CTC tie-breaks and must be reproduced.
__device__ __forceinline__ double py_round3(double x) {
    // round(x, 3) for x in [0, 1]: nearest k/1000 (ties cannot occur for non-representable
    // thousandths; exact halves x = (2k+1)/2000 are not binary fractions), then k/1000.0
    double y = __dmul_rn(x, 1000.0);
    double k = rint(y);
    // correct the (rare) case where x*1000 rounded across the .5 boundary
    double lo = __ddiv_rn(k - 0.5, 1000.0), hi = __ddiv_rn(k + 0.5, 1000.0);
    if (x < lo) k -= 1.0; else if (x > hi) k += 1.0;
    return __ddiv_rn(k, 1000.0);
}

#define CAND_PCAP 3072   // proposals before de-duplication: 1 + 3*127 + 128*20 worst case
#define CAND_HT 4096     // open-addressing table (key -> first proposal index)

// Parallel form of the ordered, de-duplicated union: (1) lay out every proposal in reference
// order, (2) a hash table keeps the FIRST proposal index of each (start, span) key, (3) an
// ordered compaction of the proposals that are their key's first occurrence.
__device__ void candidates_block(const QvTables &tab, const QvWork &wk, const QvKnobs &kn, int b, unsigned char *smem) {
    double *pscore = (double *)smem;                       // [CAND_PCAP]
    uint32_t *pkey = (uint32_t *)(pscore + CAND_PCAP);     // [CAND_PCAP]
    uint32_t *htk = pkey + CAND_PCAP;                      // [CAND_HT]
    uint32_t *hti = htk + CAND_HT;                         // [CAND_HT]
    int32_t *refs = (int32_t *)(hti + CAND_HT);            // [128]
    int32_t *roff = refs + 128;                            // [129]
    int32_t *wsum = roff + 132;                            // [8]: 4 wave sums + the leader counter
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    QvUtt &u = wk.utt[b];
    if (u.q_len == 0) return;
    const int N = tab.n_verses;
    int K = kn.top_text < QV_RUNNER_CAP ? kn.top_text : QV_RUNNER_CAP;
    if (K > N) K = N;
    const bool has_base = u.base_start >= 0;
    const int nrun = has_base ? u.n_runners : 0;
    const int nA = (has_base ? 1 : 0) + nrun + 2 * K;      // singles + base, in order
    const int32_t *ri = wk.runner_idx + (size_t)b * QV_RUNNER_CAP;
    const double *rs = wk.runner_score + (size_t)b * QV_RUNNER_CAP;
    const int32_t *si = wk.top_search + (size_t)b * QV_RUNNER_CAP;
    const double *ss = wk.top_search_sc + (size_t)b * QV_RUNNER_CAP;
    const int32_t *pi = wk.top_p3 + (size_t)b * QV_RUNNER_CAP;
    const double *ps = wk.top_p3_sc + (size_t)b * QV_RUNNER_CAP;
    for (int i = tid; i < CAND_HT; i += 256) { htk[i] = 0xFFFFFFFFu; hti[i] = 0xFFFFFFFFu; }
    // segment A + the ref list (first top_span_refs entries of the same sequence)
    const int nrefs = min(min(nA, kn.top_span_refs), 128);
    for (int i = tid; i < nA; i += 256) {
        int j = i, st, sp = 1;
        double sc;
        if (has_base && j == 0) { st = u.base_start; sp = u.base_span; sc = u.base_score; }
        else {
            j -= has_base ? 1 : 0;
            if (j < nrun) { st = ri[j]; sc = py_round3(rs[j]); }
            else if (j < nrun + K) { st = si[j - nrun]; sc = ss[j - nrun]; }
            else { st = pi[j - nrun - K]; sc = ps[j - nrun - K]; }
        }
        pkey[i] = (uint32_t)st * 8u + (uint32_t)sp;
        pscore[i] = sc;
        if (i < nrefs) refs[i] = st;
    }
    __syncthreads();
    // span expansion sizes per ref (c2c-direct/run.py:300-309)
    for (int r = tid; r < nrefs; r += 256) {
        int v = refs[r], s = tab.surah[v], a = tab.ayah[v], max_ayah = tab.surah_len[s - 1];
        int lo = a - kn.max_span + 1; if (lo < 1) lo = 1;
        int hi = a < max_ayah ? a : max_ayah, cnt = 0;
        for (int st = lo; st <= hi; ++st) {
            int e0 = a > st + 1 ? a : st + 1;
            int e1 = st + kn.max_span - 1; if (e1 > max_ayah) e1 = max_ayah;
            if (e1 >= e0) cnt += e1 - e0 + 1;
        }
        roff[r + 1] = cnt;
    }
    __syncthreads();
    if (tid == 0) {
        roff[0] = nA;
        for (int r = 0; r < nrefs; ++r) roff[r + 1] += roff[r];
    }
    __syncthreads();
    int nP = roff[nrefs];
    bool overflow = nP > CAND_PCAP;
    if (overflow) nP = CAND_PCAP;
    for (int r = tid; r < nrefs; r += 256) {
        int v = refs[r], s = tab.surah[v], a = tab.ayah[v];
        int s0 = tab.surah_start[s - 1], max_ayah = tab.surah_len[s - 1];
        int lo = a - kn.max_span + 1; if (lo < 1) lo = 1;
        int hi = a < max_ayah ? a : max_ayah, o = roff[r];
        for (int st = lo; st <= hi; ++st) {
            int e0 = a > st + 1 ? a : st + 1;
            int e1 = st + kn.max_span - 1; if (e1 > max_ayah) e1 = max_ayah;
            for (int en = e0; en <= e1; ++en, ++o)
                if (o < CAND_PCAP) { pkey[o] = (uint32_t)(s0 + st - 1) * 8u + (uint32_t)(en - st + 1); pscore[o] = 0.0; }
        }
    }
    __syncthreads();
    // first occurrence per key
    for (int i = tid; i < nP; i += 256) {
        uint32_t key = pkey[i], slot = (key * 2654435761u) >> 20;  // 12 bits
        for (;;) {
            uint32_t old = atomicCAS(&htk[slot], 0xFFFFFFFFu, key);
            if (old == 0xFFFFFFFFu || old == key) { atomicMin(&hti[slot], (uint32_t)i); break; }
            slot = (slot + 1) & (CAND_HT - 1);
        }
    }
    __syncthreads();
    // ordered compaction: thread t owns proposals [t*per, (t+1)*per)
    const int per = (nP + 255) / 256;
    int i0 = tid * per, i1 = min(nP, i0 + per), cnt = 0;
    unsigned keepmask = 0;  // per <= 12
    for (int i = i0; i < i1; ++i) {
        uint32_t key = pkey[i], slot = (key * 2654435761u) >> 20;
        while (htk[slot] != key) slot = (slot + 1) & (CAND_HT - 1);
        if (hti[slot] == (uint32_t)i) { keepmask |= 1u << (i - i0); ++cnt; }
    }
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int x = __shfl_up(incl, o);
        if (lane >= o) incl += x;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = incl - cnt;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    int32_t *cs = wk.cand_start + (size_t)b * QV_CAND_CAP;
    int32_t *cp = wk.cand_span + (size_t)b * QV_CAND_CAP;
    double *csc = wk.cand_score + (size_t)b * QV_CAND_CAP;
    for (int i = i0; i < i1; ++i) {
        if (!(keepmask >> (i - i0) & 1u)) continue;
        if (base < QV_CAND_CAP) {
            cs[base] = (int32_t)(pkey[i] >> 3); cp[base] = (int32_t)(pkey[i] & 7u); csc[base] = pscore[i];
            // the table now maps a key to its CANDIDATE index (every keepmask loop is behind the barrier above)
            uint32_t key = pkey[i], slot = (key * 2654435761u) >> 20;
            while (htk[slot] != key) slot = (slot + 1) & (CAND_HT - 1);
            hti[slot] = (uint32_t)base;
        } else {
            uint32_t key = pkey[i], slot = (key * 2654435761u) >> 20;
            while (htk[slot] != key) slot = (slot + 1) & (CAND_HT - 1);
            hti[slot] = 0xFFFFFFFFu;
        }
        ++base;
    }
    const int ncand = total < QV_CAND_CAP ? total : QV_CAND_CAP;
    if (tid == 0) {
        u.n_cand = ncand;
        if (overflow || total > QV_CAND_CAP) u.flags |= QV_FLAG_CAND_OVERFLOW;
    }
    // ---- plan of the CTC rerank: which candidates run an alpha recursion ----------------------------------
    // The alpha values of the states 0..2P of a target do not depend on what follows its first P tokens, so
    // ONE recursion over the ids of (start, span k') also yields the loss of every (start, span k < k') whose
    // ids are a prefix of them (tok_pfx): read alpha_T at the states 2P and 2P - 1.  A candidate's leader is
    // the feasible candidate with the same start verse and the largest span its ids are a prefix of; only
    // leaders run (k_ctc), each writes the losses of its members.  Same arithmetic per state as a recursion
    // of its own: the results are bit-identical, the work drops by the sharing factor (windows around a
    // reference verse share their start verse up to max_span - 1 times).
    int32_t *nlead = wsum + 4;
    if (tid == 0) *nlead = 0;
    int16_t *lead = wk.cand_lead + (size_t)b * QV_CAND_CAP;
    int16_t *memb = wk.cand_memb + (size_t)b * QV_CAND_CAP * QV_MAX_SPAN;
    float *closs = wk.cand_loss + (size_t)b * QV_CAND_CAP;
    double *cfin = wk.cand_final + (size_t)b * QV_CAND_CAP;
    for (int i = tid; i < ncand * QV_MAX_SPAN; i += 256) memb[i] = -1;
    __syncthreads();
    const int T = u.t_frames, scap = wk.t_cap > 384 ? 768 : 384;
    for (int c = tid; c < ncand; c += 256) {
        const int st = cs[c], sp = cp[c];
        const size_t k0 = (size_t)st * QV_MAX_SPAN;
        const int L = (int)(tab.tok_off[k0 + sp] - tab.tok_off[k0 + sp - 1]);
        if (!(L > 0 && 2 * L + 1 <= T && 2 * L + 1 <= scap)) {     // gate of c2c-direct/run.py:332: no loss
            closs[c] = INFINITY;
            cfin[c] = -INFINITY;
            continue;
        }
        int leader = c;
        const unsigned chain = tab.tok_pfx[st];
        for (int k = kn.max_span; k > sp; --k) {
            const unsigned need = ((1u << (k - 1)) - 1u) & ~((1u << (sp - 1)) - 1u);   // flags sp .. k - 1
            if ((chain & need) != need) continue;
            const int Lk = (int)(tab.tok_off[k0 + k] - tab.tok_off[k0 + k - 1]);
            if (!(2 * Lk + 1 <= T && 2 * Lk + 1 <= scap)) continue;
            uint32_t key = (uint32_t)st * 8u + (uint32_t)k, slot = (key * 2654435761u) >> 20;
            while (htk[slot] != key && htk[slot] != 0xFFFFFFFFu) slot = (slot + 1) & (CAND_HT - 1);
            if (htk[slot] != key || hti[slot] == 0xFFFFFFFFu) continue;
            leader = (int)hti[slot];
            break;
        }
        if (leader == c) lead[atomicAdd(nlead, 1)] = (int16_t)c;
        else memb[(size_t)leader * QV_MAX_SPAN + sp - 1] = (int16_t)c;
    }
    __syncthreads();
    // Round 6: the leaders in order of DECREASING state count (64-state buckets).  k_ctc hands leader i to wave i mod 256 of the
    // utterance: with ~400 leaders of 64 ... 448 states in arbitrary order, a wave could draw two long recursions and another two
    // short ones; sorted, the waves that take a second leader take the SHORTEST ones.  The order changes no loss.
    {
        const int nl = *nlead;
        int32_t *bcnt = (int32_t *)pscore;                 // [16] bucket counts, then [16] bucket offsets (pscore is dead by now)
        int16_t *tmp = (int16_t *)(bcnt + 32);             // [QV_CAND_CAP]
        if (tid < 32) bcnt[tid] = 0;
        __syncthreads();
        int bk[QV_CAND_CAP / 256], ps[QV_CAND_CAP / 256];
#pragma unroll
        for (int j = 0; j < QV_CAND_CAP / 256; ++j) {
            const int i = tid + j * 256;
            bk[j] = -1;
            if (i < nl) {
                const int c = lead[i], st = cs[c], sp = cp[c];
                const size_t k0 = (size_t)st * QV_MAX_SPAN;
                const int L = (int)(tab.tok_off[k0 + sp] - tab.tok_off[k0 + sp - 1]);
                bk[j] = 15 - min(15, L >> 5);             // bucket 0 = the longest targets
                ps[j] = atomicAdd(&bcnt[bk[j]], 1);
            }
        }
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int k = 0; k < 16; ++k) { bcnt[16 + k] = run; run += bcnt[k]; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < QV_CAND_CAP / 256; ++j)
            if (bk[j] >= 0) tmp[bcnt[16 + bk[j]] + ps[j]] = lead[tid + j * 256];
        __syncthreads();
        for (int i = tid; i < nl; i += 256) lead[i] = tmp[i];
        if (tid == 0) u.n_lead = nl;
    }
}

__global__ __launch_bounds__(256) void k_candidates(QvTables tab, QvWork wk, QvKnobs kn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x >= *wk.n_fail) return;
    candidates_block(tab, wk, kn, wk.fail_list[blockIdx.x], smem);
}

// ------------------------------------------------------------------ 9. CTC -------------
// ATen LossCTC.cpp float32 alpha recursion (what F.ctc_loss runs on CPU for the reference,
// c2c-direct/run.py:354-362).  One wave per target; state s = lane*NS + k.
template <int NS>
__device__ void ctc_wave(const float *__restrict__ lp, int T, const uint16_t *__restrict__ tgt, int L, int lane, float *sa) {
    // The recursion runs in log2 units (alpha2 = alpha * log2 e): v_exp_f32 / v_log_f32 are
    // base-2, so this drops four multiplies per state update; "-inf" is a large finite sentinel,
    // which removes the all-(-inf) special case (exp2(0) = 1, and the sentinel absorbs the rest).
    const float NEG = -1e30f, LOG2E = 1.44269504088896340736f;
    int S = 2 * L + 1;
    int tok[NS];
    bool skip_ok[NS];
    float a[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        int s = lane * NS + k;
        tok[k] = QV_BLANK;
        skip_ok[k] = false;
        if (s < S && (s & 1)) {
            tok[k] = tgt[s >> 1];
            skip_ok[k] = s > 1 && tgt[s >> 1] != tgt[(s >> 1) - 1];
        }
        a[k] = NEG;
        if (s == 0) a[k] = lp[QV_BLANK] * LOG2E;
        if (s == 1) a[k] = lp[tok[k]] * LOG2E;
    }
    // the gathers lp[t][tok] do not depend on the recurrence: fetch TCH frames ahead so their
    // L2 latency overlaps the exp/log chain instead of serialising with it.  (Requesting the NEXT group before this
    // one's steps -- two register sets -- was measured: 70 -> 131 VGPRs, 7 -> 3 waves per SIMD, 144 -> 190 us.  The
    // kernel has ~200 leader waves per utterance and is bound by how many of them a SIMD interleaves, not by one
    // wave's latency.)
    constexpr int TCH = NS <= 2 ? 8 : (NS <= 6 ? 4 : 2);
    for (int t0 = 1; t0 < T; t0 += TCH) {
        float lpv[TCH][NS];
#pragma unroll
        for (int j = 0; j < TCH; ++j) {
            int t = t0 + j < T ? t0 + j : T - 1;
            const float *row = lp + (size_t)t * QV_VOCAB;
#pragma unroll
            for (int k = 0; k < NS; ++k) lpv[j][k] = row[tok[k]] * LOG2E;
        }
#pragma unroll
        for (int j = 0; j < TCH; ++j) {
            if (t0 + j >= T) break;
            // previous lane's last two states
            float p1 = __shfl_up(a[NS - 1], 1), p2 = NS >= 2 ? __shfl_up(a[NS - 2], 1) : __shfl_up(a[NS - 1], 2);
            if (lane == 0) { p1 = NEG; p2 = NEG; }
            if (NS == 1 && lane == 1) p2 = NEG;
            float na[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                float la1 = a[k];
                float la2 = k >= 1 ? a[k - 1] : p1;
                float la3 = k >= 2 ? a[k - 2] : (k == 1 ? p1 : p2);
                if (NS == 1) la3 = p2;
                if (!skip_ok[k]) la3 = NEG;
                // log-sum-exp of three: the largest term is exp2(0) = 1 exactly, so only the other two need the
                // transcendental unit (the kernel is bound by v_exp_f32 / v_log_f32 issue: 4 -> 3 per state and frame)
                const float lamax = __builtin_fmaxf(la1, __builtin_fmaxf(la2, la3));
                const float lamin = __builtin_fminf(la1, __builtin_fminf(la2, la3));
                const float lamed = __builtin_fmaxf(__builtin_fminf(la1, la2), __builtin_fminf(__builtin_fmaxf(la1, la2), la3));
                // states beyond 2L never feed a lower state, so they are left unmasked
                na[k] = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(lamed - lamax) + __builtin_amdgcn_exp2f(lamin - lamax)) +
                        lamax + lpv[j][k];
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) a[k] = na[k];
        }
    }
    // alpha_T of every state, for the caller's read-outs (its own target and the prefixes it leads)
#pragma unroll
    for (int k = 0; k < NS; ++k) sa[lane * NS + k] = a[k];
}

// Single-instruction three-operand forms (the compiler pads fmaxf / fminf chains with canonicalising v_max_f32 x, x
// because a gathered log-prob could be a signalling NaN; it never is here)
__device__ __forceinline__ float f_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float f_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float f_med3(float a, float b, float c) { float r; asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float f_max2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float f_min2(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// Round 6: the same recursion with fewer instructions per state and frame -- same operands, same operation order, same
// bits as ctc_wave (tests/test_gpu_postlogits.py compares the two on every instantiation):
//  * with an even NS the parity of state lane*NS + k is the parity of k, a compile-time constant per register: even
//    states are blanks, whose third term (the skip transition) does not exist -- ctc_wave adds exp2(NEG - max) = +0 for
//    it -- so they take a two-term log-sum-exp (2 instead of 3 transcendental issues, 6 instead of ~14 VALU), and
//    their emission is the one value row[blank] per frame instead of a gather per state;
//  * max / min / median of three as v_max3 / v_min3 / v_med3 (one instruction each);
//  * states updated in place from the highest register down (a[k] needs the OLD a[k-1], a[k-2]): no copy pass.
// The kernel is bound by VALU + transcendental issue (sum over leaders of T x states, ~2.6 G state steps per batch of
// 64 gated 30 s clips), so this is where its time goes: 116 -> ~72 issue cycles per state and frame.
template <int NS>
__device__ void ctc_wave2(const float *__restrict__ lp, int T, const uint16_t *__restrict__ tgt, int L, int lane, float *sa) {
    const float NEG = -1e30f, LOG2E = 1.44269504088896340736f;
    constexpr bool PAR = (NS % 2) == 0;
    int S = 2 * L + 1;
    int tok[NS];
    bool skip_ok[NS];
    float a[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        int s = lane * NS + k;
        tok[k] = QV_BLANK;
        skip_ok[k] = false;
        if (s < S && (s & 1)) {
            tok[k] = tgt[s >> 1];
            skip_ok[k] = s > 1 && tgt[s >> 1] != tgt[(s >> 1) - 1];
        }
        a[k] = NEG;
        if (s == 0) a[k] = lp[QV_BLANK] * LOG2E;
        if (s == 1) a[k] = lp[tok[k]] * LOG2E;
    }
    // (a parity-specialised lane gathers only its NS / 2 token states + the one blank value per frame: four frames ahead fit
    // the registers up to NS = 8, where the generic program stopped at NS = 6)
    constexpr int TCH = NS <= 2 ? 8 : (NS <= (PAR ? 8 : 6) ? 4 : 2);
    for (int t0 = 1; t0 < T; t0 += TCH) {
        float lpv[TCH][NS], lpb[TCH];
#pragma unroll
        for (int j = 0; j < TCH; ++j) {
            int t = t0 + j < T ? t0 + j : T - 1;
            const float *row = lp + (size_t)t * QV_VOCAB;
            if (PAR) lpb[j] = row[QV_BLANK] * LOG2E;
#pragma unroll
            for (int k = 0; k < NS; ++k)
                if (!PAR || (k & 1)) lpv[j][k] = row[tok[k]] * LOG2E;
        }
#pragma unroll
        for (int j = 0; j < TCH; ++j) {
            if (t0 + j >= T) break;
            float p1 = __shfl_up(a[NS - 1], 1), p2 = NS >= 2 ? __shfl_up(a[NS - 2], 1) : __shfl_up(a[NS - 1], 2);
            if (lane == 0) { p1 = NEG; p2 = NEG; }
            if (NS == 1 && lane == 1) p2 = NEG;
#pragma unroll
            for (int k = NS - 1; k >= 0; --k) {
                const float la1 = a[k];
                const float la2 = k >= 1 ? a[k - 1] : p1;
                if (PAR && !(k & 1)) {
                    const float hi2 = f_max2(la1, la2), lo2 = f_min2(la1, la2);
                    a[k] = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(lo2 - hi2)) + hi2 + lpb[j];
                } else {
                    float la3 = k >= 2 ? a[k - 2] : (k == 1 ? p1 : p2);
                    if (NS == 1) la3 = p2;
                    if (!skip_ok[k]) la3 = NEG;
                    const float lamax = f_max3(la1, la2, la3), lamin = f_min3(la1, la2, la3), lamed = f_med3(la1, la2, la3);
                    a[k] = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(lamed - lamax) + __builtin_amdgcn_exp2f(lamin - lamax)) +
                           lamax + lpv[j][k];
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) sa[lane * NS + k] = a[k];
}

// loss of the target made of the first P tokens, from the alpha_T row a recursion over at least P tokens left
// in `sa` (log2 units): -ln(alpha_T(2P) + alpha_T(2P - 1)), INFINITY when no alignment exists.
__device__ __forceinline__ float ctc_readout(const float *sa, int P) {
    const float LN2 = 0.693147180559945309417f;
    const float l1 = sa[2 * P], l2 = sa[2 * P - 1];
    const float m = fmaxf(l1, l2);
    const float ll2 = __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(l1 - m) + __builtin_amdgcn_exp2f(l2 - m)) + m;
    if (ll2 < -1e29f) return INFINITY;  // no alignment (the caller applies zero_infinity)
    return -ll2 * LN2;
}

// LONG = engine capacity above 30 s (more than 384 states per target): only then are the wide
// instantiations compiled into the kernel, keeping the common kernel's register footprint small
template <bool LONG, int VAR>
__device__ void ctc_dispatch(const float *lp, int T, const uint16_t *tgt, int L, int lane, float *sa) {
    int S = 2 * L + 1;
    if (VAR == 0) {     // the wave program of rounds 1-5 (QVERSE_CTC=0 / qv_debug_kernel_variant(4, 0)): cross-check only
        if (S <= 64) return ctc_wave<1>(lp, T, tgt, L, lane, sa);
        if (S <= 128) return ctc_wave<2>(lp, T, tgt, L, lane, sa);
        if (S <= 192) return ctc_wave<3>(lp, T, tgt, L, lane, sa);
        if (S <= 256) return ctc_wave<4>(lp, T, tgt, L, lane, sa);
        if (S <= 384 || !LONG) return ctc_wave<6>(lp, T, tgt, L, lane, sa);
        if (S <= 512) return ctc_wave<8>(lp, T, tgt, L, lane, sa);
        return ctc_wave<12>(lp, T, tgt, L, lane, sa);
    }
    if (S <= 64) return ctc_wave2<1>(lp, T, tgt, L, lane, sa);
    if (S <= 128) return ctc_wave2<2>(lp, T, tgt, L, lane, sa);
    if (S <= 192) return ctc_wave2<3>(lp, T, tgt, L, lane, sa);
    if (S <= 256) return ctc_wave2<4>(lp, T, tgt, L, lane, sa);
    if (S <= 384 || !LONG) return ctc_wave2<6>(lp, T, tgt, L, lane, sa);
    if (S <= 512) return ctc_wave2<8>(lp, T, tgt, L, lane, sa);
    return ctc_wave2<12>(lp, T, tgt, L, lane, sa);
}

// One wave per LEADER candidate (k_candidates' plan): one alpha recursion, then the loss of the leader and of
// every candidate whose ids are a prefix of its ids.
// One wave per LEADER candidate (k_candidates' plan): one alpha recursion, then the loss of the leader and of
// every candidate whose ids are a prefix of its ids.
template <bool LONG, int VAR>
__global__ __launch_bounds__(256) void k_ctc(QvTables tab, QvWork wk, QvKnobs kn, const float *__restrict__ lp, int t_max) {
    __shared__ float sa_all[4][LONG ? 768 : 384];
    // grid = (utterance, block of four leader slots): the hardware dispatches workgroups x-fastest, and k_candidates sorted the
    // leaders by decreasing state count -- so EVERY utterance's longest recursions start first and the shortest ones fill the tail
    // (with the utterance on y, the last utterances' long leaders started when everything else was done)
    if ((int)blockIdx.x >= *wk.n_fail) return;
    int b = wk.fail_list[blockIdx.x];
    const QvUtt &u = wk.utt[b];
    if (!u.use_ctc) return;
    int lane = threadIdx.x & 63, wave = blockIdx.y * 4 + (threadIdx.x >> 6), nwave = gridDim.y * 4;
    float *sa = sa_all[threadIdx.x >> 6];
    int T = u.t_frames;
    const float *lpb = lp + (size_t)b * t_max * QV_VOCAB;
    const int16_t *lead = wk.cand_lead + (size_t)b * QV_CAND_CAP;
    for (int i = wave; i < u.n_lead; i += nwave) {
        const int c = lead[i];
        const int st = wk.cand_start[(size_t)b * QV_CAND_CAP + c], sp = wk.cand_span[(size_t)b * QV_CAND_CAP + c];
        const size_t k0 = (size_t)st * QV_MAX_SPAN;
        const int L = (int)(tab.tok_off[k0 + sp] - tab.tok_off[k0 + sp - 1]);
        ctc_dispatch<LONG, VAR>(lpb, T, tab.tok + tab.tok_off[k0 + sp - 1], L, lane, sa);
        // the leader itself (k == sp) and its members
        const int16_t *memb = wk.cand_memb + ((size_t)b * QV_CAND_CAP + c) * QV_MAX_SPAN;
        for (int k = 1; k <= sp; ++k) {
            const int m = k == sp ? c : (int)memb[k - 1];
            if (m < 0) continue;
            const int P = (int)(tab.tok_off[k0 + k] - tab.tok_off[k0 + k - 1]);
            float loss = ctc_readout(sa, P);
            if (isinf(loss)) loss = 0.f;  // zero_infinity=True
            float norm = __fdiv_rn(loss, (float)P);
            // -norm + TEXT_WEIGHT*text_score - SPAN_PENALTY*(span_len-1), in Python doubles
            double tw = __dmul_rn(kn.text_weight, wk.cand_score[(size_t)b * QV_CAND_CAP + m]);
            double fin = __dsub_rn(__dadd_rn(-(double)norm, tw), __dmul_rn(kn.span_penalty, (double)(k - 1)));
            if (lane == 0) {
                wk.cand_loss[(size_t)b * QV_CAND_CAP + m] = loss;
                wk.cand_final[(size_t)b * QV_CAND_CAP + m] = fin;
            }
        }
    }
}

template <int VAR>
__global__ __launch_bounds__(64) void k_ctc_debug(const float *__restrict__ lp, int T, const uint16_t *__restrict__ tg,
                                                  const int32_t *__restrict__ off, int n, float *__restrict__ loss) {
    __shared__ float sa[768];
    int c = blockIdx.x, lane = threadIdx.x;
    if (c >= n) return;
    int L = off[c + 1] - off[c];
    ctc_dispatch<true, VAR>(lp, T, tg + off[c], L, lane, sa);
    float l = ctc_readout(sa, L);
    if (lane == 0) loss[c] = l;
}

// ------------------------------------------------------------------ 10. decision -------
// mixed/run.py:96-133.  ranked = stable sort by final desc of finite-loss candidates -> the
// winner is the first maximum.
__global__ __launch_bounds__(256) void k_result(QvTables tab, QvWork wk, int batch) {
    __shared__ double sh_s[8];
    __shared__ unsigned long long sh_k[8];
    int b = blockIdx.x, tid = threadIdx.x;
    QvUtt &u = wk.utt[b];
    double best = -INFINITY;
    unsigned long long key = ~0ull;
    if (u.use_ctc) {
        for (int c = tid; c < u.n_cand; c += 256) {
            float l = wk.cand_loss[(size_t)b * QV_CAND_CAP + c];
            if (!isfinite(l)) continue;
            double f = wk.cand_final[(size_t)b * QV_CAND_CAP + c];
            if (key == ~0ull || better(f, (unsigned long long)c, best, key)) { best = f; key = c; }
        }
    }
    // block_best with -inf scores: "better" handles -inf vs -inf via key
    block_best(best, key, sh_s, sh_k);
    if (tid != 0) return;
    qv_result r;
    r.surah = r.ayah = r.ayah_end = 0;
    r.source = QV_SOURCE_NONE;
    r.score = 0.0;
    r.base_score = u.base_start >= 0 ? u.base_score : 0.0;
    r.ctc_norm_loss = 0.f;
    r.n_tokens = u.n_tok;
    r.n_chars = u.q_len;
    r.n_candidates = u.n_cand;
    r.flags = u.flags;
    r.t_frames = u.t_frames;
    if (u.q_len > 0) {
        if (u.use_ctc && key != ~0ull) {
            int c = (int)key;
            int st = wk.cand_start[(size_t)b * QV_CAND_CAP + c], sp = wk.cand_span[(size_t)b * QV_CAND_CAP + c];
            size_t tk = (size_t)st * QV_MAX_SPAN + (sp - 1);
            int L = (int)(tab.tok_off[tk + 1] - tab.tok_off[tk]);
            float norm = __fdiv_rn(wk.cand_loss[(size_t)b * QV_CAND_CAP + c], (float)L);
            r.source = QV_SOURCE_CTC;
            r.ctc_norm_loss = norm;
            r.score = exp(-(double)norm);  // host recomputes with libm for the authoritative value
            r.surah = tab.surah[st]; r.ayah = tab.ayah[st]; r.ayah_end = r.ayah + sp - 1;
            u.win = c; u.win_norm = norm;
        } else if (u.base_start >= 0) {
            r.source = QV_SOURCE_TEXT;
            r.score = u.base_score;
            r.surah = tab.surah[u.base_start]; r.ayah = tab.ayah[u.base_start];
            r.ayah_end = r.ayah + u.base_span - 1;
        }
    }
    wk.results[b] = r;
    wk.packed[b * 4 + 0] = r.surah;
    wk.packed[b * 4 + 1] = r.ayah;
    wk.packed[b * 4 + 2] = r.ayah_end;
    wk.packed[b * 4 + 3] = __float_as_int((float)r.score);
}

__global__ void k_init_utts(QvWork wk, const int32_t *__restrict__ t_dev, int batch) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) { *wk.n_fail = 0; wk.frag_ctr[0] = 0; wk.frag_ctr[1] = 0; wk.frag_ctr[2] = 0; }
    if (b < batch) wk.utt[b].t_frames = t_dev[b];
}

// ------------------------------------------------------------------ 12. verse tracker --
// VerseTracker._find_best_match (shared/verse_tracker.py:67-101) with _score_verse (:41-65)
// inlined.  grid (blocks of 256 verses, texts); the text's match masks are built in LDS by the
// block itself, then one verse per lane: LCS(text, verse) and - when the text has fewer words
// than the verse - LCS(text, first n_text words of the verse), both by streaming the verse
// through the text's bit-vector.  Scores are Python double arithmetic, operation by operation.
__global__ __launch_bounds__(256) void k_track(QvTables tab, QvTrack tw) {
    __shared__ unsigned long long spm[QV_NSYM * QV_PMS];
    __shared__ double sh_s[8];
    __shared__ unsigned long long sh_k[8];
    // grid = (text, chunk of the verse order): chunk 0 = every text's longest verses, dispatched first (see k_lcs_full)
    const int b = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    const int m = tw.meta[b * 4], n_text = tw.meta[b * 4 + 1], bonus = tw.meta[b * 4 + 2];
    const uint8_t *q = tw.q + tw.meta[b * 4 + 3];
    for (int i = tid; i < QV_NSYM * QV_PMS; i += 256) spm[i] = 0ull;
    __syncthreads();
    for (int i = tid; i < m; i += 256) { int c = q[i]; if (c < QV_NSYM) atomicOr(&spm[c * QV_PMS + (i >> 6)], 1ull << (i & 63)); }
    __syncthreads();
    const int W = (m + 63) >> 6;
    // lanes walk the verses in order of decreasing length, so the 64 verses of a wave cost about
    // the same; the first-maximum rule lives in the key (verse index), not in the visiting order
    const int slot = chunk * 256 + tid;
    const int v = slot < tab.n_verses ? tab.len_order[slot] : -1;
    double best = 0.0;
    unsigned long long bkey = ~0ull;
    if (v >= 0 && m > 0) {
        double score = 0.0;
        int matched_nb = 0;
        const int nl = tab.nobsm_len[v];
        // j: 0 = clean full, 1 = clean prefix, 2 = no_bsm full, 3 = no_bsm prefix (one LCS call site)
        int lfull = 0, n = 0, plen = 0, nwv = 0;
        const uint8_t *p = nullptr;
        for (int j = 0; j < 4; ++j) {
            if (j >= 2 && nl == 0) break;
            if ((j & 1) == 0) {
                n = j == 0 ? tab.clean_len[v] : nl;
                p = tab.clean + tab.clean_off[v] + (j == 0 ? 0 : tab.clean_len[v] - nl);
                nwv = tab.nw[j == 0 ? 0 : 2][v];
                int pw = n_text < nwv ? n_text : nwv;     // prefix = first pw words of the verse
                plen = n;
                if (pw < nwv) {
                    // the no_bsm text is a suffix of the clean text: its words are the clean text's last nwv
                    int drop = tab.nw[0][v] - nwv, shift = tab.clean_len[v] - n;
                    plen = pw > 0 ? (int)tab.wend[tab.wend_off[v] + drop + pw - 1] - shift : 0;
                }
            } else if (plen == n) continue;               // prefix is the whole verse
            int l = lcs_dispatch(W, (const uint64_t *)spm, QV_PMS, p, (j & 1) ? plen : n, m);
            if ((j & 1) == 0) { lfull = l; if (plen != n) continue; }
            double ps = ratio_from(l, m, plen), fs = ratio_from(lfull, m, n);
            double cov = __ddiv_rn((double)n_text, (double)(nwv > 1 ? nwv : 1));
            double raw = cov > 0.8 ? __dadd_rn(__dmul_rn(0.3, ps), __dmul_rn(0.7, fs))
                                   : __dadd_rn(__dmul_rn(0.7, ps), __dmul_rn(0.3, fs));
            if (v == bonus) raw = __dadd_rn(raw, 0.15);
            if (j < 2) score = raw;
            else if (raw > score) { score = raw; matched_nb = 1; }
        }
        best = score;
        bkey = (unsigned long long)v * 2ull + (unsigned long long)matched_nb;
    }
    block_best(best, bkey, sh_s, sh_k);
    if (tid == 0) {
        tw.part_s[(size_t)b * QV_TRACK_BLOCKS + chunk] = best;
        tw.part_k[(size_t)b * QV_TRACK_BLOCKS + chunk] = bkey;
    }
}

__global__ void k_track_final(QvTables tab, QvTrack tw, int batch, int nblk) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double best = 0.0;
    unsigned long long key = ~0ull;
    for (int i = 0; i < nblk; ++i) {
        double s = tw.part_s[(size_t)b * QV_TRACK_BLOCKS + i];
        unsigned long long k = tw.part_k[(size_t)b * QV_TRACK_BLOCKS + i];
        if (better(s, k, best, key)) { best = s; key = k; }
    }
    qv_track_match r;
    r.verse = -1; r.surah = 0; r.ayah = 0; r.variant = 0; r.n_words = 0; r.reserved = 0; r.score = 0.0;
    if (best > 0.0 && key != ~0ull) {          // "score > best_score" starts from 0.0 (verse_tracker.py:78,90)
        int v = (int)(key >> 1), nb = (int)(key & 1ull);
        r.verse = v; r.surah = tab.surah[v]; r.ayah = tab.ayah[v];
        r.variant = nb ? 2 : 0; r.n_words = tab.nw[nb ? 2 : 0][v]; r.score = best;
    }
    tw.out[b] = r;
}

}  // namespace

// ====================================================================== host side =======

static int launch_retrieval(qv_engine *eng, int batch, int force_ctc, hipStream_t stream) {
    QvTables &tab = eng->tab;
    QvWork &wk = eng->work;
    QvKnobs kn = eng->knobs;
    int N = tab.n_verses;
    size_t sm_tri = (size_t)N * 8 + 8 * 8 + 8 * 8 + 64 * 4 + 512 * 4 + 272 * 4 + 128 * 8 + 64 * 4 + TRI_WORDS * 4 +
                    QV_MAXQ * 2 + 8 * 4 + 64;
    hipLaunchKernelGGL(k_trigram, dim3(batch), dim3(256), sm_tri, stream, tab, wk);
    hipLaunchKernelGGL(k_lcs_full, dim3(batch, 32), dim3(256), 0, stream, tab, wk, 0);
    hipLaunchKernelGGL(k_frag, dim3(FRAG_GRID), dim3(256), 0, stream, tab, wk);
    size_t sm_p1 = (size_t)N * 8 + 128 * 8 + 128 * 4 + 272 * 4 + 128 * 4 + 128 * 8 + 64;
    hipLaunchKernelGGL(k_pass1_final, dim3(batch), dim3(256), sm_p1, stream, tab, wk, kn);
    if (qv_kernel_variant(QV_KV_SPANS) == 1) hipLaunchKernelGGL(k_spans2, dim3(QV_SPAN_BLOCKS, batch), dim3(256), 0, stream, tab, wk, kn);
    else hipLaunchKernelGGL(k_spans, dim3(QV_SPAN_BLOCKS, batch), dim3(256), 0, stream, tab, wk, kn);
    hipLaunchKernelGGL(k_base_final, dim3((batch + 63) / 64), dim3(64), 0, stream, tab, wk, kn, batch, force_ctc);
    // gate-failed utterances only (device-side list; blocks past n_fail exit at once)
    hipLaunchKernelGGL(k_lcs_full, dim3(batch, 74), dim3(256), 0, stream, tab, wk, 1);   // 3 jobs per verse: one round of 74 x 256 lanes
    hipLaunchKernelGGL(k_frag, dim3(FRAG_GRID), dim3(256), 0, stream, tab, wk);
    hipLaunchKernelGGL(k_topk, dim3(batch, 2), dim3(256), sm_p1, stream, tab, wk, kn);
    hipLaunchKernelGGL(k_candidates, dim3(batch), dim3(256), (size_t)CAND_PCAP * 12 + CAND_HT * 8 + 1200, stream, tab, wk, kn);
    return QV_OK;
}

int qv_post_run(qv_engine *eng, const float *lp, int t_max, const int32_t *t_host, int batch, hipStream_t stream) {
    QvTables &tab = eng->tab;
    QvWork &wk = eng->work;
    if (batch > wk.max_batch || t_max > wk.t_cap) {
        qv_set_error(eng, "batch or frame count exceeds engine capacity");
        return QV_ERR_CAPACITY;
    }
    for (int b = 0; b < batch; ++b)
        if (t_host[b] < 0 || t_host[b] > t_max) { qv_set_error(eng, "t_host[b] out of range"); return QV_ERR_ARG; }
    {
        // pinned staging slot: wait for the copy that last read it (two calls ago), never for the stream
        QvCtx &c = eng->ctx[eng->cur_ctx];
        const int slot = c.t_slot;
        c.t_slot = (slot + 1) % QV_STAGE_SLOTS;
        if (c.t_pending[slot]) { QV_HIP(hipEventSynchronize(c.t_copied[slot])); c.t_pending[slot] = false; }
        int32_t *th = eng->t_host_scratch + (size_t)slot * wk.max_batch;
        for (int b = 0; b < batch; ++b) th[b] = t_host[b];
        QV_HIP(hipMemcpyAsync(eng->t_dev, th, sizeof(int32_t) * batch, hipMemcpyHostToDevice, stream));
        QV_HIP(hipEventRecord(c.t_copied[slot], stream));
        c.t_pending[slot] = true;
    }
#ifdef QV_DEV_HOOKS
    static const int skip = [] { const char *e = getenv("QVERSE_SKIP"); return e ? atoi(e) : 0; }();   // dev-only, see qv_model.hip
#else
    constexpr int skip = 0;
#endif
    auto launch_chain = [&]() -> int {
        if (skip & 128) return QV_OK;   // (timing experiments only: no post-logits chain at all)
        hipLaunchKernelGGL(k_decode, dim3(batch), dim3(64 * DEC_WAVES), 0, stream, tab, wk, lp, t_max, eng->t_dev);
        qv_stage_mark(eng, 2, stream);
        int rc = launch_retrieval(eng, batch, 0, stream);
        if (rc) return rc;
        qv_stage_mark(eng, 3, stream);
        if (skip & 16) { }
        // the long-target variant (up to 768 states per candidate: 12 state registers per lane, 3 KB of LDS per wave) only when
        // THIS batch has a clip of more than 384 frames -- not whenever the engine COULD hold one: an engine created for
        // 30 s clips used to run every 10 s batch through it (round 4: tools/sweep.py's 10 s row, 4.8 ms, against bench.py's
        // 3.8 ms for the same batch on an engine sized for it).  2L + 1 <= T <= t_max <= 384 holds for every candidate then.
        else if (qv_kernel_variant(QV_KV_CTC) == 0) {
            if (t_max > 384) hipLaunchKernelGGL((k_ctc<true, 0>), dim3(batch, 64), dim3(256), 0, stream, tab, wk, eng->knobs, lp, t_max);
            else hipLaunchKernelGGL((k_ctc<false, 0>), dim3(batch, 64), dim3(256), 0, stream, tab, wk, eng->knobs, lp, t_max);
        }
        else if (t_max > 384) hipLaunchKernelGGL((k_ctc<true, 1>), dim3(batch, 64), dim3(256), 0, stream, tab, wk, eng->knobs, lp, t_max);
        else hipLaunchKernelGGL((k_ctc<false, 1>), dim3(batch, 64), dim3(256), 0, stream, tab, wk, eng->knobs, lp, t_max);
        hipLaunchKernelGGL(k_result, dim3(batch), dim3(256), 0, stream, tab, wk, batch);
        qv_stage_mark(eng, 4, stream);
        return QV_OK;
    };
    // One graph launch for the whole chain when its arguments are the ones a graph was captured with: the
    // engine's own log-prob workspace (qv_predict_batch*), same batch and frame count -- the steady state of a
    // serving loop.  Everything the 13 kernels decide per utterance (gate, candidate counts, leaders) is read
    // from device memory, so a replay is the same work as the launches it was captured from.  Caller-owned
    // log-prob tensors (pointer changes per call), profiled runs and single-context engines (the chain then runs
    // on the CALLER's stream, which may be the legacy default stream -- not capturable) take the plain launches.
    // Opt-in (QVERSE_POST_GRAPH=1): measured on one MI355X with three batches in flight it changes nothing
    // (4.26 vs 4.13 ms per step, within box noise) -- kernel boundaries cost the same inside a graph, and the
    // host's ~3.5 us per launch is not what limits the step.
    static const bool use_graph = [] { const char *e = getenv("QVERSE_POST_GRAPH"); return e && atoi(e) != 0; }();
    QvCtx &gc = eng->ctx[eng->cur_ctx];
    if (use_graph && !gc.post_graph_off && eng->n_ctx > 1 && stream == gc.stream && lp == eng->logprobs_ws && !eng->profile_stages) {
        // (the kernel variants that pick launches inside the chain are part of the key: a graph captured under another span pass
        // or CTC wave program must not be replayed after qv_debug_kernel_variant changed it)
        const int variants = qv_kernel_variant(QV_KV_SPANS) | (qv_kernel_variant(QV_KV_CTC) << 4);
        QvCtx::PostGraph *hit = nullptr;
        for (int i = 0; i < gc.n_post_graph; ++i)
            if (gc.post_graph[i].lp == lp && gc.post_graph[i].batch == batch && gc.post_graph[i].t_max == t_max && gc.post_graph[i].variants == variants)
                hit = &gc.post_graph[i];
        bool ran_plain = false;
        if (!hit && gc.n_post_graph < 4) {
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            // as for the forward graph: a failure of the capture machinery is not a failure of the batch
            hipError_t e0 = hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal), e1 = hipSuccess, e2 = hipSuccess;
            if (e0 == hipSuccess) {
                int rc = launch_chain();
                e1 = hipStreamEndCapture(stream, &graph);
                if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
                if (e1 == hipSuccess) e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                if (graph) (void)hipGraphDestroy(graph);
            }
            if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess || !exec) {
                (void)hipGetLastError();
                gc.post_graph_off = true;
                int rc = launch_chain();
                if (rc) return rc;
                ran_plain = true;
            } else {
                gc.post_graph[gc.n_post_graph] = {lp, batch, t_max, variants, exec};
                hit = &gc.post_graph[gc.n_post_graph++];
            }
        }
        if (hit) {
            QV_HIP(hipGraphLaunch(hit->exec, stream));
        } else if (!ran_plain) {
            int rc = launch_chain();
            if (rc) return rc;
        }
    } else {
        int rc = launch_chain();
        if (rc) return rc;
    }
    QV_HIP(hipGetLastError());
    eng->last_batch = batch;
    eng->last_tmax = t_max;
    return QV_OK;
}

int qv_post_debug_retrieve(qv_engine *eng, const uint8_t *codes_host, int n, hipStream_t stream) {
    QvWork &wk = eng->work;
    if (n > QV_MAXQ) { qv_set_error(eng, "transcript longer than QV_MAX_TRANSCRIPT"); return QV_ERR_CAPACITY; }
    int32_t zero = 0;
    QV_HIP(hipMemcpyAsync(eng->t_dev, &zero, sizeof(int32_t), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_init_utts, dim3(1), dim3(64), 0, stream, wk, eng->t_dev, 1);
    if (n > 0) QV_HIP(hipMemcpyAsync(wk.q, codes_host, n, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_prepare_codes, dim3(1), dim3(64), 0, stream, wk, n);
    int rc = launch_retrieval(eng, 1, 1, stream);
    if (rc) return rc;
    QV_HIP(hipGetLastError());
    QV_HIP(hipStreamSynchronize(stream));
    return QV_OK;
}

int qv_post_debug_ctc(qv_engine *eng, const float *lp, int T, const uint16_t *tg, const int32_t *lens, int n,
                      float *loss_host, hipStream_t stream) {
    std::vector<int32_t> off(n + 1, 0);
    for (int i = 0; i < n; ++i) {
        if (lens[i] <= 0 || 2 * lens[i] + 1 > 768) { qv_set_error(eng, "target length unsupported"); return QV_ERR_ARG; }
        off[i + 1] = off[i] + lens[i];
    }
    uint16_t *d_t = nullptr;
    int32_t *d_o = nullptr;
    float *d_l = nullptr;
    QV_HIP(hipMalloc(&d_t, sizeof(uint16_t) * std::max(1, off[n])));
    QV_HIP(hipMalloc(&d_o, sizeof(int32_t) * (n + 1)));
    QV_HIP(hipMalloc(&d_l, sizeof(float) * std::max(1, n)));
    QV_HIP(hipMemcpyAsync(d_t, tg, sizeof(uint16_t) * off[n], hipMemcpyHostToDevice, stream));
    QV_HIP(hipMemcpyAsync(d_o, off.data(), sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice, stream));
    if (qv_kernel_variant(QV_KV_CTC) == 0) hipLaunchKernelGGL(k_ctc_debug<0>, dim3(n), dim3(64), 0, stream, lp, T, d_t, d_o, n, d_l);
    else hipLaunchKernelGGL(k_ctc_debug<1>, dim3(n), dim3(64), 0, stream, lp, T, d_t, d_o, n, d_l);
    QV_HIP(hipMemcpyAsync(loss_host, d_l, sizeof(float) * n, hipMemcpyDeviceToHost, stream));
    QV_HIP(hipStreamSynchronize(stream));
    (void)hipFree(d_t); (void)hipFree(d_o); (void)hipFree(d_l);
    return QV_OK;
}

int qv_post_tracker_match(qv_engine *eng, const uint8_t *codes_host, const int32_t *offsets_host,
                          const int32_t *n_words_host, const int32_t *bonus_host, int batch,
                          qv_track_match *out_host, hipStream_t stream) {
    QvTrack &tw = eng->track;
    const int nblk = (eng->tab.n_verses + 255) / 256;
    if (nblk > QV_TRACK_BLOCKS) { qv_set_error(eng, "verse table larger than the tracker workspace"); return QV_ERR_CAPACITY; }
    std::vector<int32_t> meta((size_t)QV_TRACK_CAP * 4);
    for (int b0 = 0; b0 < batch; b0 += QV_TRACK_CAP) {
        int nb = std::min(QV_TRACK_CAP, batch - b0);
        const int32_t base = offsets_host[b0];
        for (int i = 0; i < nb; ++i) {
            meta[i * 4] = offsets_host[b0 + i + 1] - offsets_host[b0 + i];
            meta[i * 4 + 1] = n_words_host[b0 + i];
            meta[i * 4 + 2] = bonus_host[b0 + i];
            meta[i * 4 + 3] = offsets_host[b0 + i] - base;
        }
        const int32_t total = offsets_host[b0 + nb] - base;   // <= nb * QV_MAXQ (lengths were checked)
        if (total > 0) QV_HIP(hipMemcpyAsync(tw.q, codes_host + base, (size_t)total, hipMemcpyHostToDevice, stream));
        QV_HIP(hipMemcpyAsync(tw.meta, meta.data(), sizeof(int32_t) * 4 * nb, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_track, dim3(nb, nblk), dim3(256), 0, stream, eng->tab, tw);
        hipLaunchKernelGGL(k_track_final, dim3((nb + 63) / 64), dim3(64), 0, stream, eng->tab, tw, nb, nblk);
        QV_HIP(hipGetLastError());
        QV_HIP(hipMemcpyAsync(out_host + b0, tw.out, sizeof(qv_track_match) * nb, hipMemcpyDeviceToHost, stream));
        QV_HIP(hipStreamSynchronize(stream));   // meta[] and the workspace are reused by the next slice
    }
    return QV_OK;
}

// match_verse(text, max_span, hint, use_trigram_index=False) (quran_db.py:244-371): full scan,
// continuation bonuses, span pass; result in utt[0].base_* (SYNCHRONOUS).
int qv_post_match_verse(qv_engine *eng, const uint8_t *codes_host, int n, int n_bonus, const int32_t *bonus_verse,
                        const double *bonus_value, int max_span, hipStream_t stream) {
    QvTables &tab = eng->tab;
    QvWork &wk = eng->work;
    QvKnobs kn = eng->knobs;
    kn.max_span = max_span;
    const int N = tab.n_verses;
    int32_t zero = 0;
    QV_HIP(hipMemcpyAsync(eng->t_dev, &zero, sizeof(int32_t), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_init_utts, dim3(1), dim3(64), 0, stream, wk, eng->t_dev, 1);
    if (n > 0) QV_HIP(hipMemcpyAsync(wk.q, codes_host, n, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_prepare_codes, dim3(1), dim3(64), 0, stream, wk, n);
    int hv[3] = {-1, -1, -1};
    double hb[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < n_bonus; ++i) { hv[i] = bonus_verse[i]; hb[i] = bonus_value[i]; }
    hipLaunchKernelGGL(k_set_hint, dim3(1), dim3(1), 0, stream, wk, 0, n_bonus, hv[0], hv[1], hv[2], hb[0], hb[1], hb[2]);
    size_t sm_tri = (size_t)N * 8 + 8 * 8 + 8 * 8 + 64 * 4 + 512 * 4 + 272 * 4 + 128 * 8 + 64 * 4 + TRI_WORDS * 4 +
                    QV_MAXQ * 2 + 8 * 4 + 64;
    hipLaunchKernelGGL(k_trigram, dim3(1), dim3(256), sm_tri, stream, tab, wk);
    hipLaunchKernelGGL(k_lcs_full, dim3(1, 74), dim3(256), 0, stream, tab, wk, 0);
    hipLaunchKernelGGL(k_frag, dim3(FRAG_GRID), dim3(256), 0, stream, tab, wk);
    if (n_bonus > 0) hipLaunchKernelGGL(k_hint_sp, dim3(1), dim3(64), 0, stream, tab, wk, 0);
    size_t sm_p1 = (size_t)N * 8 + 128 * 8 + 128 * 4 + 272 * 4 + 128 * 4 + 128 * 8 + 64;
    hipLaunchKernelGGL(k_pass1_final, dim3(1), dim3(256), sm_p1, stream, tab, wk, kn);
    if (qv_kernel_variant(QV_KV_SPANS) == 1) hipLaunchKernelGGL(k_spans2, dim3(QV_SPAN_BLOCKS, 1), dim3(256), 0, stream, tab, wk, kn);
    else hipLaunchKernelGGL(k_spans, dim3(QV_SPAN_BLOCKS, 1), dim3(256), 0, stream, tab, wk, kn);
    hipLaunchKernelGGL(k_base_final, dim3(1), dim3(64), 0, stream, tab, wk, kn, 1, 0);
    QV_HIP(hipGetLastError());
    QV_HIP(hipStreamSynchronize(stream));
    return QV_OK;
}
