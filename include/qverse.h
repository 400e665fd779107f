/*
 * qverse.h -- C ABI of libqverse.so, the MI355X-native replacement for the numerics of
 * the reference's c2c-direct-mixed hot path (paths relative to yazinsai/offline-tarteel):
 *
 *   audio_signal f32[B,N] --qv_forward--> log-probs f32[B,T,1025]
 *        replaces  onnxruntime InferenceSession.run(None, {"audio_signal","length"})
 *                  experiments/c2c-direct-mixed/run.py:55-63, .../c2c-direct-mixed-tta/run.py:74-79
 *   log-probs --qv_decode_retrieve_rerank--> (surah, ayah, ayah_end, score, source)
 *        replaces  _greedy_decode        experiments/c2c-direct/run.py:187-204
 *                  _build_candidates     experiments/c2c-direct/run.py:251-311
 *                  QuranDB.match_verse / search  shared/quran_db.py:92-99,244-371
 *                  _ctc_rerank (torch F.ctc_loss) experiments/c2c-direct/run.py:314-380
 *                  decision logic        experiments/c2c-direct-mixed/run.py:96-133
 *   qv_predict_batch = both, back to back on one stream (what predict() does per file,
 *        experiments/c2c-direct-mixed/run.py:66-133, for a whole batch).
 *
 * Conventions
 *   - plain C, no torch types.  Pointers named *_dev are DEVICE pointers owned by the
 *     caller (e.g. PyTorch-ROCm tensors); the library never frees or retains them.
 *     Pointers named *_host are host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Entry points
 *     enqueue work on it; only functions documented as synchronous wait for it.
 *   - every function returns 0 on success or a QV_ERR_* code; qv_last_error() gives text.
 *     The Python binding turns non-zero into an exception, so the runner's per-sample
 *     try/except (benchmark/runner.py:322-325) yields the reference's empty prediction.
 *   - one engine per process per GPU.  Host calls on one engine are serialised by the library (a
 *     per-engine lock held for the duration of each call), so calling from several threads is safe --
 *     the reference's TTA plugin runs its 0.9x / 1.1x passes from two threads
 *     (c2c-direct-mixed-tta/run.py:129-130) -- but gains nothing: batching the work into one call
 *     (what plugin.predict_tta does) or keeping batches in flight (qv_predict_batch_async) is the
 *     GPU-native form.  Results of a context must be fetched before that context is reused.
 *     The ASYNC protocol (qv_predict_batch_async -> qv_last_context -> qv_fetch_results_ctx) is one logical sequence:
 *     drive it from ONE thread per engine (or under the caller's own lock) -- another thread's call in between moves
 *     qv_last_context().  qv_last_error() returns a copy private to the calling thread.
 */
#ifndef QVERSE_H
#define QVERSE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QV_VOCAB 1025
#define QV_BLANK 1024
#define QV_MAX_TRANSCRIPT 1024 /* normalised transcript chars handled on device */

enum {
    QV_OK = 0,
    QV_ERR_ARG = 1,        /* bad argument / unsupported knob */
    QV_ERR_IO = 2,         /* tables or weights file missing / malformed (FileNotFoundError) */
    QV_ERR_HIP = 3,        /* HIP runtime error */
    QV_ERR_CAPACITY = 4,   /* batch / length exceeds what the engine was created for */
    QV_ERR_NO_MODEL = 5    /* forward requested but engine has no weights */
};

enum { QV_SOURCE_NONE = 0, QV_SOURCE_TEXT = 1, QV_SOURCE_CTC = 2 };

/* flags in qv_result.flags */
enum {
    QV_FLAG_EMPTY_TRANSCRIPT = 1,   /* greedy decode produced nothing -> _empty("") */
    QV_FLAG_TRANSCRIPT_TRUNCATED = 2,/* > QV_MAX_TRANSCRIPT chars: prediction withheld (surah 0) */
    QV_FLAG_USED_CTC = 4,           /* gate failed (base.score < threshold): rerank ran */
    QV_FLAG_CAND_OVERFLOW = 8       /* candidate list clipped at engine capacity */
};

typedef struct qv_engine qv_engine;

/* Arithmetic of the acoustic model.
 *   QV_PREC_FP16             f16 weights and GEMM operands, f32 accumulation / residual stream.
 *   QV_PREC_MIXED_INT4_INT8  storage formats of the reference's file with f16 arithmetic: block-128 int4 Linear weights and
 *                            per-channel int8 pointwise-conv weights, dequantised in the MFMA operand fetch (W4A16 / W8A16).
 *   QV_PREC_ORT_MIXED        the ARITHMETIC onnxruntime runs on the reference's file (experiments/c2c-direct-mixed/run.py:1-9):
 *                            MatMulNBits int4 on the Linear layers, and on EVERY Conv DynamicQuantizeLinear -> ConvInteger:
 *                            per-utterance uint8 activations (range from the tensor's own min / max), one symmetric int8
 *                            scale per weight tensor, int32 accumulation (i8 MFMA for the GEMM-shaped convolutions, exact
 *                            integer stencils for the depthwise / strided ones), float32 rescale.  csrc/qv_ort.h. */
enum { QV_PREC_FP16 = 0, QV_PREC_MIXED_INT4_INT8 = 1, QV_PREC_ORT_MIXED = 2 };

typedef struct {
    int32_t struct_size;        /* sizeof(qv_config), for forward compatibility */
    int32_t device;             /* HIP device ordinal */
    const char *tables_path;    /* qverse_tables.bin (tools/build_tables.py) */
    const char *weights_path;   /* flat weight file (tools/convert_weights.py) or NULL */
    uint64_t random_weights_seed; /* used when weights_path == NULL and with_model != 0 */
    int32_t with_model;         /* 0: post-logits stages only */
    int32_t precision;          /* QV_PREC_* */
    int32_t max_batch;          /* capacity: utterances per call */
    int32_t max_samples;        /* capacity: samples per utterance (480000 = 30 s); at most 976,000 (61 s):
                                   the CTC rerank holds 2L+1 <= T <= 768 states per candidate */
    /* CTC_DIRECT_* knobs, experiments/c2c-direct/run.py:62-74 (same defaults) */
    int32_t top_text;           /* CTC_DIRECT_TOP_TEXT        100 */
    int32_t top_span_refs;      /* CTC_DIRECT_TOP_SPAN_REFS    80 */
    int32_t max_span;           /* CTC_DIRECT_MAX_SPAN          6 (table limit 6) */
    double threshold;           /* CTC_DIRECT_THRESHOLD       0.80 */
    double text_weight;         /* CTC_DIRECT_TEXT_WEIGHT      0.0 (final = -norm_loss + weight * text score - penalty);
                                   any finite value, negative included, as c2c-direct/run.py:67 */
    double span_penalty;        /* CTC_DIRECT_SPAN_PENALTY     0.5 */
    int32_t skip_unused_passes; /* 1: skip search()/pass-3 when the gate passes (their output is
                                   unused by the mixed plugin); 0: literal */
    int32_t n_contexts;         /* 1..8 batches in flight (default 1).  With > 1,
                                   qv_predict_batch_async() rotates over that many execution
                                   contexts (own activations, workspace and internal stream), so
                                   the latency-bound decode/retrieval/CTC kernels of one batch run
                                   under the forward pass of the next.  Memory scales with it. */
} qv_config;

/* One prediction; mirrors the dict of experiments/c2c-direct-mixed/run.py:126-133. */
typedef struct {
    int32_t surah, ayah, ayah_end;  /* 0,0,0 = no match (_empty) */
    int32_t source;                 /* QV_SOURCE_* */
    double score;                   /* unrounded; the mixed plugin rounds to 4 dp, TTA does not */
    double base_score;              /* match_verse score (0 if none) */
    float ctc_norm_loss;            /* winner's loss / len when source == CTC */
    int32_t n_tokens;               /* greedy token count */
    int32_t n_chars;                /* normalised transcript length */
    int32_t n_candidates;           /* candidates scored by the rerank (0 if gate passed) */
    int32_t flags;                  /* QV_FLAG_* */
    int32_t t_frames;               /* encoder frames of this utterance */
} qv_result;

void qv_config_default(qv_config *cfg);
int qv_create(const qv_config *cfg, qv_engine **out);
void qv_destroy(qv_engine *e);
const char *qv_last_error(const qv_engine *e); /* e may be NULL: last create() error */

/* Encoder frames produced for n_samples of 16 kHz audio (three stride-2 stages over
 * floor(n/160)+1 mel frames). */
int32_t qv_frames_for_samples(int64_t n_samples);

/* Acoustic model.  audio_dev: f32[B, n_max] row-major, rows zero-padded past lengths_host[b].
 * logprobs_dev: f32[B, t_max, 1025] with t_max >= qv_frames_for_samples(max length); rows
 * t >= T[b] are ZEROED (onnxruntime hands the reference exactly [1, T, 1025], mixed/run.py:59-63: a padded batch tensor
 * never exposes uninitialised memory).  t_out_host[b] receives T[b] (computed on the host, no sync). */
int qv_forward(qv_engine *e, const float *audio_dev, const int64_t *lengths_host, int32_t batch,
               int64_t n_max, float *logprobs_dev, int32_t t_max, int32_t *t_out_host, void *stream);

/* Post-logits stages on log-probs already in HBM.  t_host[b] = valid frames of row b.
 * results_host: qv_result[B]; greedy_ids_host (optional, may be NULL): i32[B, t_max] collapsed
 * token ids (-1 padded) for host-side transcript text.  SYNCHRONOUS: returns after the results
 * have been copied back (one stream sync at the end, none in between). */
int qv_decode_retrieve_rerank(qv_engine *e, const float *logprobs_dev, const int32_t *t_host,
                              int32_t batch, int32_t t_max, qv_result *results_host,
                              int32_t *greedy_ids_host, void *stream);

/* Asynchronous variant: results stay in an engine-owned device buffer; fetch with
 * qv_fetch_results() after the stream (or an event) has completed. */
int qv_decode_retrieve_rerank_async(qv_engine *e, const float *logprobs_dev, const int32_t *t_host,
                                    int32_t batch, int32_t t_max, void *stream);
int qv_fetch_results(qv_engine *e, int32_t batch, int32_t t_max, qv_result *results_host,
                     int32_t *greedy_ids_host, void *stream);

/* forward + post-logits on the engine's own log-prob workspace.  SYNCHRONOUS like above. */
int qv_predict_batch(qv_engine *e, const float *audio_dev, const int64_t *lengths_host,
                     int32_t batch, int64_t n_max, qv_result *results_host,
                     int32_t *greedy_ids_host, void *stream);
int qv_predict_batch_async(qv_engine *e, const float *audio_dev, const int64_t *lengths_host,
                           int32_t batch, int64_t n_max, void *stream);

/* Same, and returns the execution context the batch runs on in *ctx_out under the same lock hold -- what a caller
 * that may share the engine with other threads passes to qv_wait_ctx / qv_fetch_results_ctx / qv_packed_results_ctx
 * (qv_last_context() after a separate call can already name another thread's batch). */
int qv_predict_batch_async_ctx(qv_engine *e, const float *audio_dev, const int64_t *lengths_host,
                               int32_t batch, int64_t n_max, void *stream, int32_t *ctx_out);

/* ---- a15: speed perturbation / sample-rate conversion ------------------------------------
 * The float32 polyphase FIR behind scipy.signal.resample_poly(x, up, down), which the reference's
 * TTA wrapper calls with (9, 10) and (11, 10) (experiments/c2c-direct-mixed-tta/run.py:60-71):
 *   y[m] = sum_j x[xi - (P-1) + j] * h[t + up * (P-1-j)],  xi = (m0+m)*down / up, t = (m0+m)*down % up,
 * P = ceil(n_taps / up), terms outside [0, n_in) skipped, accumulated IN THIS ORDER in float32
 * (no FMA), which is what scipy's upfirdn does -- results are bit-identical to it.
 * taps_host: n_taps float32 filter taps exactly as resample_poly hands them to upfirdn (designed,
 * scaled by `up`, zero-padded in front/behind); m0 = first kept output sample (n_pre_remove).
 * x_dev [n_in], y_dev [n_out] device pointers.  Asynchronous on `stream`. */
int qv_upfirdn(qv_engine *e, const float *x_dev, int64_t n_in, const float *taps_host, int32_t n_taps,
               int32_t up, int32_t down, int64_t m0, int64_t n_out, float *y_dev, void *stream);

/* The same FIR over a BATCH of rows in one launch: output row r = resample of source row src_rows_host[r] (NULL: row r) of
 * x_dev [.., x_pitch], n_in_host[r] samples long; y_dev row r [y_pitch] receives ceil(n_in * up / down) samples followed by
 * zeros up to y_pitch -- the engine's zero-padded [B, N] input layout, so the result can go straight into qv_predict_batch.
 * Used by the TTA wrapper (all 0.9x or all 1.1x copies of a batch's gated clips: one launch instead of one per clip) and by
 * the device ingest of non-16 kHz files (160/441, 1/3: offline-tarteel_amd/audio.py load_audio_device; reference
 * shared/audio.py:8-18).  Per sample the arithmetic is qv_upfirdn's, term for term.  rows <= 1024.  Asynchronous on
 * `stream`; the host arrays may be released on return. */
int qv_upfirdn_batch(qv_engine *e, const float *x_dev, int64_t x_pitch, const int32_t *src_rows_host, const int64_t *n_in_host,
                     int32_t rows, const float *taps_host, int32_t n_taps, int32_t up, int32_t down, int64_t m0,
                     float *y_dev, int64_t y_pitch, void *stream);
/* Interleaved multi-channel rows [frames][channels] (float32) -> mono rows: the float32 mean over the channel axis (sequential
 * sum, one division), i.e. numpy's x.reshape(-1, ch).mean(axis=1) -- the reference's mix-down (shared/audio.py:13-15). */
int qv_mixdown_batch(qv_engine *e, const float *x_dev, int64_t x_pitch, const int64_t *n_frames_host, int32_t rows,
                     int32_t channels, float *y_dev, int64_t y_pitch, void *stream);

/* ---- batches in flight (n_contexts > 1) -------------------------------------------------
 * qv_predict_batch_async() then only ORDERS ITS INPUTS on `stream` (the audio must stay unchanged
 * until the call's results have been joined) and runs on the context's internal stream; a call
 * blocks the host only when the context it is about to reuse is still busy.  Results are joined
 * per context: */
int32_t qv_context_count(const qv_engine *e);      /* may be lower than qv_config.n_contexts, see below */
/* How many of FOUR fresh HIP streams -- what a four-context engine is about to create -- the runtime runs side by side on
 * this device: 4, 2 or 1 (0 = probe failed).  Four one-wave spin kernels on four streams, elapsed time over spin time
 * (~1 ms, cached per process).  The runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the
 * variable ONCE, when HIP initialises; streams sharing a queue serialise.  Measured: with 8 queues 7 fresh streams run
 * concurrently (4 once RCCL holds its own), on the default 4 queues only 3.  qv_create() calls this when n_contexts >= 4
 * and falls back to 3 contexts -- the best measured setting on the default queues -- unless all four ran concurrently, so
 * that a host which touched HIP before exporting GPU_MAX_HW_QUEUES=8 loses ~2 % instead of ~14 %. */
int32_t qv_probe_concurrent_streams(void);
/* dev tool: the probe's raw figure (elapsed time / spin time) for n_streams fresh streams, not cached */
double qv_debug_probe_rounds(int32_t n_streams);
int32_t qv_last_context(const qv_engine *e);     /* context used by the most recent async call */
/* Host-side join: blocks the calling thread until context `ctx`'s last batch has finished (no-op for an idle
 * context).  A device-side join (qv_packed_results_ctx on a stream) parks a wait on an OLDER batch in that stream's
 * hardware queue; when the runtime maps another stream onto the same queue, that stream's work is stuck behind it. */
int qv_wait_ctx(qv_engine *e, int32_t ctx);
/* makes `stream` wait for that context's batch, then returns its packed i32[B,4] rows */
const int32_t *qv_packed_results_ctx(qv_engine *e, int32_t ctx, void *stream);
/* host copy of that context's results (SYNCHRONOUS) */
int qv_fetch_results_ctx(qv_engine *e, int32_t ctx, int32_t batch, int32_t t_max, qv_result *results_host,
                         int32_t *greedy_ids_host);

/* ---- streaming row: the verse tracker's matching step -------------------------------------
 * Replaces VerseTracker._find_best_match (shared/verse_tracker.py:67-101) with _score_verse
 * (:41-65) inlined: for each of `batch` accumulated texts, the scan of all 6,236 verses (clean
 * text, and the bismillah-stripped variant where the verse has one) of
 *     raw = blend(ratio(text, first min(n_text, n_verse) words of the verse), ratio(text, verse))
 *           [+ 0.15 for the verse after the last emission]
 * and the first maximum in verse order.  Texts arrive as alphabet codes (0 = ' ', 63 = a
 * character outside the verse alphabet), concatenated, text b = codes[offsets[b] ..
 * offsets[b+1]); n_words_host[b] = number of whitespace-separated words of text b;
 * bonus_verse_host[b] = global index of the verse that gets the continuation bonus
 * (QuranDB.get_next_verse of the last emission, shared/quran_db.py:81-90) or -1.
 * out_host[b].verse = -1 when no verse scores above 0.0; the minimum-emit-score and
 * minimum-word-count gates of the caller are NOT applied here.  Texts longer than
 * QV_MAX_TRANSCRIPT codes are refused (QV_ERR_CAPACITY).  SYNCHRONOUS on `stream`. */
typedef struct qv_track_match {
    int32_t verse;     /* global verse index, -1 = none */
    int32_t surah, ayah;
    int32_t variant;   /* 0 = text_clean matched, 2 = text_clean_no_bsm */
    int32_t n_words;   /* words of the matched text (what _emit trims by, verse_tracker.py:110-114) */
    int32_t reserved;
    double score;
} qv_track_match;
int qv_tracker_match(qv_engine *e, const uint8_t *codes_host, const int32_t *offsets_host,
                     const int32_t *n_words_host, const int32_t *bonus_verse_host, int32_t batch,
                     qv_track_match *out_host, void *stream);

/* QuranDB.match_verse(text, max_span, hint) WITHOUT the trigram restriction (shared/quran_db.py:
 * 244-371 with use_trigram_index=False), as StreamingPipeline.run_on_full_transcript calls it
 * (shared/streaming.py:86): pass 1 scores every verse (fragment scores of text_clean /
 * text_clean_alt / text_clean_no_bsm; for the <= 3 verses of the continuation hint also
 * _suffix_prefix_score, :188-208, plus their bonus, _continuation_bonuses :121-146), pass 2 every
 * window of 2..max_span ayat of the surahs of the top 20.  codes_host: the NORMALISED text as
 * alphabet codes; bonus_verse / bonus_value: n_bonus (0..3) global verse indices and their
 * bonuses; max_span in [2, 8].  Outputs the winner (first verse index, number of ayat, score);
 * the caller applies the threshold.  SYNCHRONOUS on `stream`.  Uses the workspace of the most
 * recently used execution context (waits for batches in flight first): fetch the results of an
 * asynchronous batch before calling it. */
int qv_match_verse(qv_engine *e, const uint8_t *codes_host, int32_t n_codes, int32_t n_bonus,
                   const int32_t *bonus_verse, const double *bonus_value, int32_t max_span,
                   int32_t *start, int32_t *span, double *score, void *stream);

/* Device pointer of the packed (surah, ayah, ayah_end, float-bits(score)) i32[B,4] rows of the
 * last async call -- the payload of the per-batch RCCL all-gather. */
const int32_t *qv_packed_results_dev(qv_engine *e);

/* ---- stage-level entry points used by the parity tests (each SYNCHRONOUS) ------------- */

/* match_verse/search/pass-3/_build_candidates for one already-normalised transcript given
 * as alphabet codes.  Outputs: base (start verse index, span, score); candidate list in
 * reference order as (start index, span, text score).  cand_cap = capacity of the arrays. */
int qv_debug_retrieve(qv_engine *e, const uint8_t *codes_host, int32_t n_codes,
                      int32_t *base_start, int32_t *base_span, double *base_score,
                      int32_t *cand_start, int32_t *cand_span, double *cand_score,
                      int32_t cand_cap, int32_t *n_cand, int32_t *runner_idx, double *runner_score,
                      int32_t *n_runners, void *stream);

/* CTC negative log-likelihood (float32 alpha recursion) of n target sequences against one
 * [T,1025] log-prob matrix in HBM.  targets_host: concatenated u16 ids, lens_host[n]. */
int qv_debug_ctc_loss(qv_engine *e, const float *logprobs_dev, int32_t t_frames,
                      const uint16_t *targets_host, const int32_t *lens_host, int32_t n,
                      float *loss_host, void *stream);

/* Intermediate activations of the acoustic model for the layer-wise parity tests.
 * what: 0 = normalised mel features f32[B, t_mel_max, 80]; 1 = subsampling output
 * f32[B, t_max, 512]; 2 = encoder output after layer `layer` f32[B, t_max, 512].
 * QV_PREC_ORT_MIXED only (the tensors in front of / behind each quantiser of the conv path):
 * 3 = norm_conv output, 4 = GLU output, 5 = depthwise conv + BatchNorm + Swish output of layer `layer`,
 * each f32[B, t_max, 512]; 6 = conv.2 output f32[B, t2_max, 20, 256], 7 = ReLU(conv.3) (same shape),
 * 8 = conv.5 output f32[B, t_max, 10, 256], 9 = ReLU(conv.6) f32[B, t_max, 10, 256] (rows past an
 * utterance's length are unspecified for 6..9). */
int qv_debug_forward_tap(qv_engine *e, int32_t what, int32_t layer, float *out_dev, void *stream);

/* Measurement hooks for bench.py's roofline line: while enabled, every GEMM launch of the
 * acoustic model is bracketed by HIP events on its own stream.  After synchronising the stream,
 * qv_profile_gemm_read() returns, per kernel class c = epilogue*3 + tile (0 = 64-wide, 1 = 128-wide,
 * 2 = 256 x 256; 21 classes), the summed event time in ms, the summed algorithmic FLOPs (2*M*N*K)
 * and the launch count, and clears the log. */
int qv_profile_gemm(qv_engine *e, int32_t enable);
int qv_profile_gemm_read(qv_engine *e, double *ms21, double *flops21, int32_t *launches21);
/* Replays ONE GEMM of layer 0 with the shapes of the last forward `iters` times back to back
 * between two HIP events on `stream` (SYNCHRONOUS).  which: 0 FFN-up [M,512]x[512,2048]+Swish,
 * 1 FFN-down [M,2048]x[2048,512]+residual, 2 QKV, 3 attention out-projection, 4 pointwise-conv+GLU.
 * Returns the average launch duration in microseconds and the algorithmic FLOPs (2*M*N*K). */
int qv_profile_replay_gemm(qv_engine *e, int32_t which, int32_t iters, double *avg_us, double *flops_per_launch,
                           void *stream);
/* Name of the kernel that replay runs, i.e. the tile shape the launcher picks for that GEMM at the last
 * forward's row count ("k_gemm256<f16_swish>", "k_gemm<resid,128>", ...). */
int qv_profile_replay_kernel(qv_engine *e, int32_t which, char *name_out, int32_t name_cap);
/* Process-wide GEMM tile policy, for the tests that compare tile shapes bit for bit: 0 = 128-wide tiles only,
 * 1 = the default (256 x 256 tiles where N % 256 == 0 and the grid has >= 160 of them), 2 = 256 x 256 wherever
 * the shape allows, -1 = back to the environment (QVERSE_GEMM_T256) / default.  The tile shape never changes
 * a result: both kernels form the same products in the same accumulation order. */
int qv_debug_gemm_tiles(int32_t mode);
/* Tile HEIGHT of the 256-wide kernel: 0 = 256 rows always, 1 = the default (192-row tiles where they save a round of tiles
 * over the 256 CUs and fewer than three batches are in flight), 2 = that rule whatever is in flight, 3 = 192 rows wherever
 * the wide kernel runs, -1 = back to the environment (QVERSE_GEMM_BM) / default.  Same products, same accumulation order:
 * the tile height never changes a result either.  Both switches bump an epoch that is part of the forward-graph key. */
int qv_debug_gemm_tile_height(int32_t mode);
/* Process-wide attention kernel variant, for the tests: 3 = the default: an utterance of at most 128 encoder frames
 * (10.2 s) is served by the single-pass short-utterance kernel, a longer one by the key-tiled kernel -- by its OWN length,
 * so the bits of an utterance never depend on the batch it travels in; 0 = the key-tiled kernel (two heads per block)
 * for every utterance, 1 = one head per block, 2 = one wave per query tile, 4 = k_attention_x (key-tiled, four self-staging
 * waves per (head, 128-query group), two blocks per CU) for every utterance, 5 = the short-utterance kernel + k_attention_x;
 * -1 = back to the environment (QVERSE_ATT_X=1 / QVERSE_ATT_TILED=1 / QVERSE_ATT_HPB=1 / QVERSE_ATT_OLD=1, read once per
 * process) / default.  Variants 0, 1, 2, 4 give identical bits, so do 3 and 5; 3 / 5 differ from the others for short
 * utterances by the softmax's summation order only (one pass over the row instead of a running maximum). */
int qv_debug_attention_variant(int32_t mode);

/* Process-wide variant of a single kernel, for the tests that compare two implementations of one stage bit for bit
 * (-1 = back to the environment / default).  which 0 (QVERSE_LOGMEL): the log-mel kernel's 256-point FFT -- 0 = Stockham
 * through LDS, 1 = in registers with DPP / v_permlane*_swap exchanges; which 1 (QVERSE_ORT_SUB): conv.0 of
 * QV_PREC_ORT_MIXED's front end -- 0 = VALU, 1 = v_mfma_f32_32x32x2_f32 on the integer-valued operands; which 2
 * (QVERSE_SPANS): match_verse's span pass -- 0 = one LCS walk per span, 1 = one walk per start verse with the count read
 * off at every ayah end; which 3 (QVERSE_FWD_GRAPH): the forward of an engine with more than one context -- 0 = plain
 * launches, 1 = a shape that repeats on a context is captured once and replayed as one hipGraph launch; which 4
 * (QVERSE_CTC): the alpha recursion of the CTC rerank -- 0 = the wave program of rounds 1-5, 1 = the parity-specialised
 * one (two-term log-sum-exp for blank states).  The variants of a kernel produce identical bits. */
int qv_debug_kernel_variant(int32_t which, int32_t mode);

/* How many forwards of this engine were replayed as a hipGraph launch, and how many graphs were captured, since creation
 * (tests / bench.py: shows that the replay path -- and not the plain launches -- is what ran). */
int qv_debug_forward_graph_stats(qv_engine *eng, int64_t *replays, int64_t *captures);
/* Captures / graph instantiations that failed since creation (a capture-unsafe call of the host invalidated one, ...): the
 * batch is then issued as plain launches and the context stops capturing -- never an error of the batch.  -1: no model. */
int64_t qv_debug_forward_graph_failures(qv_engine *eng);

/* Measurement hook for bench.py's `realistic_mix` leg.  Seeded random weights decode every synthetic clip to a near-empty
 * transcript, so the headline workload never sees a recitation the text match recognises.  While log-probs are injected,
 * qv_predict_batch_async() still runs the WHOLE forward pass on the audio it is given, but its post-logits stages read the
 * caller's tensor (f32[batch, t_max, 1025] in HBM, t_host[b] valid frames; must stay alive and unchanged) instead of the
 * forward's output -- e.g. verse-shaped log-probs at the v1 corpus' gate pass / fail ratio.  NULL clears the hook.  Results
 * are then predictions for the INJECTED log-probs; never use it outside measurements. */
int qv_profile_inject_logprobs(qv_engine *e, const float *logprobs_dev, int32_t t_max, const int32_t *t_host, int32_t batch);

/* Stage timers -- the device-side counterpart of C2C_DIRECT_MIXED_PROFILE (experiments/c2c-direct-mixed/
 * run.py:34,76-81,117-124: forward= decode= build= rerank= per file).  While enabled, every batch brackets
 * its four stages with HIP events on the stream it runs on: forward (acoustic model), decode (argmax +
 * greedy collapse + normalisation), build (match_verse / search / pass 3 / candidate assembly), rerank
 * (CTC losses + decision).  qv_stage_times() waits for context `ctx`'s last batch (SYNCHRONOUS) and
 * returns the four durations in milliseconds; a stage that did not run reports 0. */
int qv_profile_stages(qv_engine *e, int32_t enable);
int qv_stage_times(qv_engine *e, int32_t ctx, float *ms4_host);

/* Host-only: block-128 symmetric int4 quantisation of one Linear weight w[N][K] (f32 row-major,
 * N % 64 == 0, K % 128 == 0) followed by the inverse of the device packing, i.e. the f32 matrix
 * (q - 8) * half(scale) the W4A16 GEMM multiplies by under QV_PREC_MIXED_INT4_INT8.  Needs no GPU;
 * the parity tests compare it with the oracle's quantiser. */
int qv_debug_int4_roundtrip(const float *w, int32_t n_rows, int32_t k, float *out);
/* Same for the pointwise-convolution weights of that precision mode: per-output-channel symmetric int8
 * (N % 64 == 0, K % 64 == 0), i.e. the f32 matrix q * scale[n] the W8A16 GEMM multiplies by. */
int qv_debug_int8_roundtrip(const float *w, int32_t n_rows, int32_t k, float *out);

/* Host-only (no GPU needed): the float32 tensors a weight file must hold, in file order -- names are
 * the NeMo state-dict keys of the CTC branch -- and tensor `index` of the seeded synthetic
 * initialisation an engine created with weights_path == NULL uses.  tools/convert_weights.py is
 * written against these, so the converter and the engine cannot disagree about names or shapes. */
int32_t qv_weight_count(void);
int qv_weight_spec(int32_t index, char *name_out, int32_t name_cap, int32_t *dims4_out, int32_t *ndim_out);
int qv_weight_random(uint64_t seed, int32_t index, float *out, int64_t numel);

/* What the engine's acoustic model actually runs on, as text: the precision mode and where the quantisation grids came
 * from -- "weights quantised by the engine", or for a file converted from the reference's quantised ONNX
 * (tools/convert_weights.py --onnx marks it) how many Linear tensors sit on the FILE's own MatMulNBits grid and how many
 * run as dequantised f16 values (block sizes other than 128).  bench.py / the plugin report it next to a number. */
int qv_weights_info(qv_engine *e, char *out, int32_t cap);

/* Library build info: "gfx950;hip-x.y;..." */
const char *qv_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* QVERSE_H */
