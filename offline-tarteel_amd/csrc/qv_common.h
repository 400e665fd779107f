// qv_common.h -- internal declarations shared by the translation units of libqverse.so.
// gfx950 only; no compatibility layers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/qverse.h"

#define QV_NSYM 40          // alphabet codes 0..39 (0 = ' '); 63 = matches nothing
#define QV_OTHER 63
#define QV_MAXQ QV_MAX_TRANSCRIPT
#define QV_MAXW (QV_MAXQ / 64)
#define QV_MAX_SPAN 6
#define QV_CAND_CAP 2048
#define QV_RUNNER_CAP 128
#define QV_RAW_CAP 4096     // raw (pre-collapse) transcript code units per utterance
#define QV_MAX_VW 11        // ceil(684 / 64): words of the longest verse text

#define QV_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            qv_set_error(eng, std::string(#expr) + ": " + hipGetErrorString(e_));            \
            return QV_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

// word counts the LCS core is instantiated for; verse match-mask rows are padded to these
__host__ __device__ inline int qv_tmpl_w(int w) {
    return w <= 1 ? 1 : w <= 2 ? 2 : w <= 3 ? 3 : w <= 4 ? 4 : w <= 6 ? 6 : w <= 8 ? 8 : w <= 11 ? 11 : 16;
}

struct qv_engine;
void qv_set_error(qv_engine *e, const std::string &msg);

// ------------------------------------------------------------------ static tables -----
// Device-resident verse database (built once at qv_create from qverse_tables.bin).
struct QvTables {
    int n_verses, n_surah, n_tri, n_text;  // n_text = 2N + #nobsm
    // verse identity
    const uint8_t *surah;        // [N]
    const uint16_t *ayah;        // [N]
    const int32_t *surah_start;  // [n_surah+1]
    const int32_t *surah_len;    // [n_surah]
    // clean texts laid out back to back WITH one ' ' between consecutive verses, so every
    // span text (first.no_bsm||clean + ' ' + rest.clean) is one contiguous slice
    const uint8_t *clean;        // codes
    const uint32_t *clean_off;   // [N]
    const uint16_t *clean_len;   // [N]
    const uint16_t *nobsm_len;   // [N] 0 = none; text = last nobsm_len codes of clean[v]
    // the same texts once more for the prefix-shared span pass (k_spans2): verse v is the block [clean8_off[v],
    // clean8_off[v + 1]) = one ' ', its codes, then filler codes (0xFF: match nothing, leave the LCS recurrence alone) up
    // to a multiple of 8 -- every ayah END falls on the end of an 8-code chunk.  Built at qv_create.
    const uint8_t *clean8;
    const uint32_t *clean8_off;  // [N+1], multiples of 8
    const uint8_t *alt;
    const uint32_t *alt_off;     // [N]
    const uint16_t *alt_len;     // [N]
    const uint16_t *nw[3];       // word counts clean / alt / nobsm
    const int32_t *nobsm_rank;   // [N] index among verses that have a no_bsm text, -1 otherwise
    // word ends of the clean texts: wend[wend_off[v] + k] = chars of the first k+1 words joined
    const uint32_t *wend_off;    // [N+1]
    const uint16_t *wend;
    const int32_t *len_order;    // [N] verse indices, longest clean text first (lanes of a wave get similar lengths)
    // per-text bit-parallel match masks (pattern = the verse text): text id = variant*N + v for
    // variant 0/1, 2N + nobsm_rank for variant 2.  layout [QV_NSYM][words]
    const uint64_t *pmv;
    const uint32_t *pmv_off;     // [n_text] offset in u64 units
    // trigram index (forward form)
    const uint32_t *tri_keys;    // [n_tri] sorted packed trigrams
    const double *tri_idf;       // [n_tri]
    const uint32_t *vtri_off;    // [N+1]
    const uint16_t *vtri;        // sorted trigram ids per verse
    // ... and the inverted form: verses of each trigram in ascending order; tri_slice[id*5 + w] is
    // where the verses of quarter w of the verse range start (w = 4: end of the list)
    const uint16_t *tri_post;
    const uint16_t *tri_map;     // [2^18] packed trigram (6 bits per code) -> id, 0xFFFF = not in the index
    const uint32_t *tri_slice;   // [n_tri * 5]
    // CTC token table: key = v*6 + (span-1)
    const uint32_t *tok_off;     // [N*6+1]
    const uint16_t *tok;
    // bit k-1 of tok_pfx[v]: the ids of (v, span k) are a PREFIX of the ids of (v, span k+1) (SentencePiece
    // segments word by word, so a span's ids normally extend the shorter span's; not so where the first ayah
    // loses its bismillah in a span).  Built at qv_create; lets k_ctc share one alpha recursion between them.
    const uint8_t *tok_pfx;      // [N]
    // vocabulary pieces: normalised code strings
    const uint32_t *piece_off;   // [1026]
    const uint8_t *piece_codes;
};

// ------------------------------------------------------------------ per-batch state ---
struct QvUtt {           // one per utterance, device memory (SoA would not buy anything here)
    int32_t t_frames;
    int32_t n_tok;
    int32_t q_len, qs_len, q_words;
    int32_t flags;
    int32_t n_cand1;       // pass-1 iteration list length
    int32_t full_scan;     // 1 if pass 1 iterates all verses
    int32_t best1_idx;     // best single verse (index) after pass 1
    double best1_score;
    int32_t n_runners;
    int32_t n_surah20;
    int32_t surah20[20];
    int32_t base_start, base_span;   // -1 = none
    double base_score;
    int32_t use_ctc;
    int32_t n_cand;
    int32_t n_lead;        // candidates that run an alpha recursion (k_candidates' plan for k_ctc)
    int32_t win;           // winning candidate index or -1
    float win_norm;
    // match_verse with a continuation hint and without the trigram restriction (qv_match_verse;
    // the hot path leaves both at 0): up to 3 verses get a bonus and a suffix-prefix score
    int32_t force_full;
    int32_t hint_n;
    int32_t hint_v[3];
    double hint_bonus[3];
    double hint_sp[3];
};

struct QvWork {
    int max_batch, t_cap;
    QvUtt *utt;              // [B]
    int16_t *frame_ids;      // [B][t_cap]
    int32_t *greedy;         // [B][t_cap]
    uint8_t *q;              // [B][QV_MAXQ]
    uint8_t *qs;             // [B][QV_MAXQ] spaceless
    uint64_t *pm;            // [B][2][QV_NSYM][QV_MAXW]  (0: q, 1: spaceless q)
    int32_t *cand1;          // [B][N]
    int16_t *lcsf;           // [B][N][3] full-string LCS(transcript, text) (clean, alt, nobsm)
    double *fs;              // [B][N][3] fragment scores (clean, alt, nobsm)
    int16_t *lcs_p3;         // [B][N]   pass 3: LCS(spaceless transcript, clean text) (k_lcs_full mode 1)
    uint32_t *frag_list;     // [B * N * 3] texts whose fragment score needs the window scan: utterance << 15 | verse << 2 | variant
    int32_t *frag_ctr;       // [4]: expensive entries (front of frag_list), work-stealing cursor of k_frag, cheap entries (back of the list)
    int frag_cap;            // entries frag_list holds
    double *search_sc;       // [B][N]   search score (max over clean/alt)
    int32_t *runner_idx;     // [B][QV_RUNNER_CAP]
    double *runner_score;    // [B][QV_RUNNER_CAP]
    int32_t *top_search;     // [B][QV_RUNNER_CAP]
    double *top_search_sc;
    int32_t *top_p3;         // [B][QV_RUNNER_CAP]
    double *top_p3_sc;
    double *span_part_score; // [B][QV_SPAN_BLOCKS]
    uint64_t *span_part_key; // [B][QV_SPAN_BLOCKS]
    int32_t *cand_start;     // [B][QV_CAND_CAP]
    int32_t *cand_span;      // [B][QV_CAND_CAP]
    double *cand_score;      // [B][QV_CAND_CAP]
    int16_t *cand_lead;      // [B][QV_CAND_CAP] list of the leaders' candidate indices (first n_lead entries)
    int16_t *cand_memb;      // [B][QV_CAND_CAP][QV_MAX_SPAN] per leader: candidate index of its (start, span k) prefix, -1 = none
    float *cand_loss;        // [B][QV_CAND_CAP]
    double *cand_final;      // [B][QV_CAND_CAP]
    qv_result *results;      // [B]
    int32_t *packed;         // [B][4]
    int32_t *fail_list;      // [B] compacted gate-fail utterance indices
    int32_t *n_fail;         // [1]
};

#define QV_SPAN_BLOCKS 64

struct QvKnobs {
    int top_text, top_span_refs, max_span;
    double threshold, text_weight, span_penalty;
    int skip_unused;
};

// verse tracker workspace (qv_tracker_match): QV_TRACK_CAP texts per launch
#define QV_TRACK_CAP 256
#define QV_TRACK_BLOCKS 32   // >= ceil(n_verses / 256) partial maxima per text
#define QV_RESAMPLE_ROWS 1024   // rows per qv_upfirdn_batch / qv_mixdown_batch call
#define QV_RESAMPLE_SLOTS 16    // calls whose row tables may be in flight on a stream at once
struct QvTrack {
    uint8_t *q;          // [CAP * QV_MAXQ] codes of the slice's texts, back to back
    int32_t *meta;       // [CAP][4] q_len, n_words, bonus verse, offset in q
    double *part_s;      // [CAP][QV_TRACK_BLOCKS]
    uint64_t *part_k;    // [CAP][QV_TRACK_BLOCKS] verse * 2 + (no_bsm variant matched)
    qv_track_match *out; // [CAP]
};

// post-logits launcher (qv_postlogits.hip)
int qv_post_tracker_match(qv_engine *eng, const uint8_t *codes_host, const int32_t *offsets_host,
                          const int32_t *n_words_host, const int32_t *bonus_host, int batch,
                          qv_track_match *out_host, hipStream_t stream);
int qv_post_run(qv_engine *eng, const float *logprobs_dev, int t_max, const int32_t *t_host, int batch,
                hipStream_t stream);
int qv_post_debug_retrieve(qv_engine *eng, const uint8_t *codes_host, int n, hipStream_t stream);
int qv_post_match_verse(qv_engine *eng, const uint8_t *codes_host, int n, int n_bonus, const int32_t *bonus_verse,
                        const double *bonus_value, int max_span, hipStream_t stream);
int qv_post_debug_ctc(qv_engine *eng, const float *lp, int T, const uint16_t *tg, const int32_t *lens, int n,
                      float *loss_host, hipStream_t stream);

// acoustic model (qv_model.hip)
struct QvModel;
int qv_model_create(qv_engine *eng, const qv_config *cfg, QvModel **out);
void qv_model_destroy(QvModel *m);
int qv_model_forward(qv_engine *eng, QvModel *m, const float *audio_dev, const int64_t *len_host, int batch,
                     int64_t n_max, float *logprobs_dev, int t_max, int32_t *t_out_host, hipStream_t stream,
                     bool zero_pad_rows = false,    // true: rows t >= T[b] of logprobs_dev are zeroed (the public qv_forward)
                     bool may_graph = false);       // true: `stream` is one of the engine's own (capturable) context streams
int qv_model_tap(qv_engine *eng, QvModel *m, int what, int layer, float *out_dev, hipStream_t stream);
int qv_model_replay_gemm(qv_engine *eng, QvModel *m, int which, int iters, double *avg_us, double *flops, hipStream_t s);
int qv_model_replay_kernel(qv_engine *eng, QvModel *m, int which, char *name_out, int cap);

void qv_model_weights_info(const QvModel *m, char *out, int cap);
void qv_model_graph_stats(const QvModel *m, int64_t *replays, int64_t *captures);
int64_t qv_model_graph_failures(const QvModel *m);   // captures / instantiations that failed (the context then runs plain launches)
void qv_model_select_ctx(QvModel *m, int k);
// records stage event `i` of the current context on `s` when stage profiling is on (qv_capi.hip)
void qv_stage_mark(qv_engine *eng, int i, hipStream_t s);

// One execution context = everything a batch in flight owns (activations live in QvModel).
// With n_ctx > 1, qv_predict_batch_async() round-robins the contexts, each on its own internal
// stream, so the latency-bound post-logits kernels of one batch run under the forward of the next.
#define QV_MAX_CTX 8
// Pinned staging buffers (frame counts here, the model's length table in QvActs) are written by the host
// and then read by an asynchronous H2D copy: each has QV_STAGE_SLOTS slots used in turn, and a slot is
// reused only after the event recorded behind its copy has completed -- back-to-back asynchronous calls
// on one context (n_contexts = 1 included) never overwrite lengths a queued copy has yet to read.
#define QV_STAGE_SLOTS 2
struct QvCtx {
    QvWork work;
    float *logprobs_ws;
    int32_t *t_host_scratch;   // [QV_STAGE_SLOTS][max_batch] pinned
    int32_t *t_dev;
    hipStream_t stream;
    hipEvent_t in_ready, done;
    bool busy;
    int last_batch, last_tmax;
    hipEvent_t t_copied[QV_STAGE_SLOTS];
    bool t_pending[QV_STAGE_SLOTS];
    int t_slot;
    // stage timers (qv_profile_stages): start, forward done, decode done, build done, rerank done
    hipEvent_t stage_ev[5];
    bool stage_valid;
    // the post-logits chain (k_decode .. k_result, 13 kernels) as ONE hipGraph launch, keyed by what the kernel
    // arguments depend on; captured the first time a key is seen on this context
    struct PostGraph { const float *lp; int batch, t_max, variants; hipGraphExec_t exec; } post_graph[4];   // variants: the kernel variants in force (spans, CTC)
    bool post_graph_off = false;   // a capture / instantiate failed on this context: plain launches from then on
    int n_post_graph;
};

struct qv_engine {
    // every entry point that touches the engine takes this lock for the duration of the HOST call (entry points call
    // one another, hence recursive): calls from several host threads are serialised instead of corrupting the shared
    // host-side state (current context, staging slots); the device work they enqueue stays asynchronous
    std::recursive_mutex mu;
    // ... and consecutive calls that enqueue on DIFFERENT caller streams are ordered on the device as well (they share
    // the current context's workspace): each such call waits for the event the previous one left behind (QvStreamOrder)
    hipEvent_t tail_ev = nullptr;
    hipStream_t tail_stream = nullptr;
    bool tail_valid = false;
    qv_config cfg;
    QvKnobs knobs;
    int device;
    std::string last_error;
    QvTables tab;                 // device pointers
    std::vector<void *> allocs;   // everything hipMalloc'ed for tables/work
    QvCtx ctx[QV_MAX_CTX];
    int n_ctx, cur_ctx, next_ctx;
    // the CURRENT context's buffers (copied from ctx[cur_ctx] by qv_select_ctx; launches capture
    // pointer values, and all enqueueing is host-serial)
    QvWork work;
    QvModel *model;
    float *logprobs_ws;           // [max_batch][t_cap][1025] engine-owned log-prob workspace
    int32_t *t_host_scratch;      // pinned [max_batch]
    int32_t *t_dev;               // [max_batch]
    // resampler filters already arranged per phase and resident in HBM (qv_upfirdn)
    struct Fir { int up; std::vector<float> taps; float *hflip_dev; int P; };
    std::vector<Fir> firs;
    unsigned char *resample_tab = nullptr;   // ring of row tables for qv_upfirdn_batch / qv_mixdown_batch
    unsigned resample_at = 0;
    QvTrack track;
    bool profile_stages;
    // measurement hook (qv_profile_inject_logprobs): caller-owned log-probs the post-logits stages of
    // qv_predict_batch_async() read INSTEAD of the forward's own output (the forward still runs in full)
    const float *inject_lp;
    int inject_tmax, inject_batch;
    std::vector<int32_t> inject_t;
    // host copies of small table parts used by debug/entry code
    std::vector<uint8_t> h_surah;
    std::vector<uint16_t> h_ayah;
    int last_batch, last_tmax;
};
