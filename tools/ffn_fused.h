// qv_ffn.h -- the fused Conformer feed-forward module: out += alpha * (W2 * swish(W1 * x + b1) + b2), one kernel, the
// [M, 2048] hidden activation never leaves the CU (reference workload = InferenceSession.run,
// experiments/c2c-direct-mixed/run.py:55-63).  Layouts and the kernel: qv_ffn.hip.
#pragma once

#include "../offline-tarteel_amd/csrc/qv_kernels.h"

#define QV_FFN_UNIT_BYTES 16384          // one staged unit of weights = 16 MFMA fragments of 1 KB
#define QV_FFN_UNITS (2 * (QV_FF / 32) * 2)   // 256 units = W1 + W2 = 4 MB

struct FfnArgs {
    const half_t *X;        // [M][ldx] f16: the LayerNorm output (A operand of the first Linear)
    int ldx;
    const uint8_t *Wp;      // packed weight stream, QV_FFN_UNITS x 16 KB (qv_ffn_pack)
    const float *b1;        // [2048]
    const float *b2;        // [512]
    float *out;             // [M][ldo] f32 residual stream, read-modify-write
    int ldo, M;
    float alpha;            // 0.5 (Macaron half-step)
};

// Host side: W1 [2048][512], W2 [512][2048] (f32, row-major = the state-dict tensors) -> the unit stream the kernel walks
// front to back.  Stream order: W1(0)a W1(0)b, then for c = 0..63: [W1(c+1)a W1(c+1)b if c < 63] W2(c)a W2(c)b, where
// chunk c is hidden channels 32c..32c+31, a / b are the two K halves of W1 (256 wide) resp. the two 16-deep k-steps of W2
// (all 512 output columns each).
void qv_ffn_pack(const float *w1, const float *w2, half_t *stream_out);
void launch_ffn_fused(const FfnArgs &a, hipStream_t s);
