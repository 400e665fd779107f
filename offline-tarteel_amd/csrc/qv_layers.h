// qv_layers.h -- launchers of the non-GEMM kernels (qv_layers.hip).
#pragma once

#include "qv_kernels.h"

struct FrontendTab {
    const float *window;    // [512] symmetric Hann(400) zero-padded to n_fft, centred
    const float2 *twiddle;  // [256] exp(-2 pi i m / 512)
    const int32_t *mel_lo;  // [80] first FFT bin of each filter
    const int32_t *mel_cnt; // [80] number of bins (<= 32)
    const float *mel_w;     // [32][80]: tap-major (tap k of filter m at k * 80 + m), zero beyond a filter's mel_cnt
};

size_t qv_melstats_doubles(size_t max_batch);   // size of the `stats` buffer launch_logmel needs (results + partial sums)
void launch_logmel(const float *audio, int64_t n_max, const int32_t *n_samples, const FrontendTab &ft, float *feats,
                   int tm_max, double *stats, int batch, hipStream_t s);
void launch_melapply(const float *feats, const int32_t *n_samples, int tm_max, const double *stats, float *out, int batch,
                     hipStream_t s);
void launch_conv0(const float *feats, int tm_max, const int32_t *len_in, const double *stats, const float *w,
                  const float *bias, half_t *out, int t1_max, int batch, hipStream_t s);
void launch_sub01(const float *feats, int tm_max, const int32_t *len_mel, const double *stats, const float *w0,
                  const float *b0, const int32_t *len1, const float *w1, const float *b1, half_t *out, int t2_max, int batch,
                  hipStream_t s);
void launch_dwconv2d(const half_t *in, int tin_max, int fin, const int32_t *len_in, const float *w, const float *bias,
                     half_t *out, int tout_max, int fout, int batch, hipStream_t s);
void launch_pack_rows(const half_t *x, int t_max, int row_elems, const int32_t *len, const int32_t *row_off, half_t *y,
                      int32_t *row_map, int batch, hipStream_t s);
void launch_layernorm(const float *x, const float *g, const float *b, half_t *y, int M, hipStream_t s);
void launch_layernorm2(float *x, const float *g1, const float *b1, const float *g2, const float *b2, half_t *y, int M,
                       hipStream_t s);
void launch_to_half(const float *x, half_t *y, size_t n, hipStream_t s);
void launch_to_float(const half_t *x, float *y, size_t n, hipStream_t s);
void launch_attention(const half_t *qk, const half_t *vt, const half_t *pos, int pos_ld, const float *bu, const float *bv,
                      const int32_t *len, const int32_t *row_off, half_t *out, int t_max, int t_min, int t_pad, int batch,
                      hipStream_t s, int variant = -1);
int qv_attention_variant();   // the process-wide variant as launch_attention would read it now
// -1 environment / default (= 3); 0 two heads per block for every utterance, 1 one head per block, 2 one wave per query tile
// (0..2 bit-identical); 3 = utterances of <= 128 frames on k_attention_short, longer ones as 0
void qv_attention_set_variant(int mode);
void launch_dwconv1d(const half_t *x, const float *w, const float *bias, const int32_t *len, const int32_t *row_off, half_t *y,
                     int t_max, int batch, hipStream_t s);
void launch_logsoftmax(const float *logits, int ld, float *out, int M, const int32_t *row_map, int t_out, hipStream_t s);
// zeros rows [len[b], t_out) of every utterance of a dense [B][t_out][1025] tensor; t_min = the shortest utterance's frames
void launch_zero_pad_rows(float *out, const int32_t *len, int t_out, int t_min, int batch, hipStream_t s);
void launch_upfirdn_rows(const float *x, int64_t x_pitch, const int32_t *src, const int64_t *n_in_rows, int rows, const float *hflip,
                         int P, int up, int down, int64_t m0, float *y, int64_t y_pitch, hipStream_t s);
void launch_mixdown(const float *x, int64_t x_pitch, const int64_t *n_frames, int rows, int channels, float *y, int64_t y_pitch,
                    int64_t max_frames, hipStream_t s);
void launch_upfirdn(const float *x, int64_t n_in, const float *hflip, int P, int up, int down, int64_t m0, int64_t n_out,
                    float *y, hipStream_t s);
