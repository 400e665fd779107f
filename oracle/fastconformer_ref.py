"""fp32 PyTorch-CPU restatement of the acoustic model (row a1 of SURVEY.md section 8).

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py header).

PARITY UNPINNED for this stage: the reference runs an external ONNX file
(data/onnx_export/fastconformer_full_mixed.onnx, listed in .MISSING_LARGE_BLOBS:11) through
onnxruntime 1.24.2 (uv.lock:2858); neither the file nor onnxruntime/NeMo exists in the build
container, and none of the reference's tests hold vectors at this boundary.
This module therefore restates the PUBLIC NeMo definition of
nvidia/stt_ar_fastconformer_hybrid_large_pcd_v1.0's CTC branch (nemo-toolkit 2.7.0,
uv.lock:2313; model id at experiments/c2c-direct/run.py:50; call order preprocessor ->
encoder -> ctc_decoder at experiments/c2c-direct/run.py:179-181):

  AudioToMelSpectrogramPreprocessor / FilterbankFeatures (eval): pre-emphasis 0.97, STFT
    n_fft 512 / hop 160 / win 400 symmetric Hann, center + reflect pad, power spectrum,
    80 Slaney-normalised Slaney-scale mel filters 0-8000 Hz, log(x + 2^-24), per-feature
    mean / unbiased-std normalisation over valid frames (+1e-5), padded frames zeroed.
    In-repo anchors: docs/plans/2026-03-01-onnx-browser-migration-plan.md:174,586-594.
  ConformerEncoder (FastConformer-Large): dw_striding x8 subsampling (256 ch), xscaling,
    17 x [1/2 FFN, rel-pos MHSA (8 x 64, untied pos_bias_u/v, rel_shift), conv module
    (pointwise, GLU, depthwise k=9, BatchNorm, Swish, pointwise), 1/2 FFN, LayerNorm].
  ConvASRDecoder: 1x1 Conv1d 512 -> 1025, log_softmax.  blank = 1024
    (web/frontend/public/export_metadata.json:13-14), output already log-softmaxed (PLAN.md:96).

The HIP forward is checked against THIS module on seeded random weights (self-consistency);
real-weights parity needs the weight file supplied out of band (tools/convert_weights.py).
"""

from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

SR = 16000
N_FFT = 512
HOP = 160
WIN = 400
N_MELS = 80
D_MODEL = 512
N_HEADS = 8
D_K = 64
N_LAYERS = 17
FF = 2048
CONV_K = 9
SUB_CH = 256
VOCAB = 1025
PREEMPH = 0.97
LOG_GUARD = 2.0 ** -24


# ------------------------------------------------------------------- weights ---------
def weight_shapes(n_layers: int = N_LAYERS) -> dict[str, tuple]:
    """NeMo state-dict names (CTC branch only) -> shapes."""
    s = {
        "encoder.pre_encode.conv.0.weight": (SUB_CH, 1, 3, 3), "encoder.pre_encode.conv.0.bias": (SUB_CH,),
        "encoder.pre_encode.conv.2.weight": (SUB_CH, 1, 3, 3), "encoder.pre_encode.conv.2.bias": (SUB_CH,),
        "encoder.pre_encode.conv.3.weight": (SUB_CH, SUB_CH, 1, 1), "encoder.pre_encode.conv.3.bias": (SUB_CH,),
        "encoder.pre_encode.conv.5.weight": (SUB_CH, 1, 3, 3), "encoder.pre_encode.conv.5.bias": (SUB_CH,),
        "encoder.pre_encode.conv.6.weight": (SUB_CH, SUB_CH, 1, 1), "encoder.pre_encode.conv.6.bias": (SUB_CH,),
        "encoder.pre_encode.out.weight": (D_MODEL, SUB_CH * 10), "encoder.pre_encode.out.bias": (D_MODEL,),
        "ctc_decoder.decoder_layers.0.weight": (VOCAB, D_MODEL, 1), "ctc_decoder.decoder_layers.0.bias": (VOCAB,),
    }
    for i in range(n_layers):
        p = f"encoder.layers.{i}."
        for ln in ("norm_feed_forward1", "norm_self_att", "norm_conv", "norm_feed_forward2", "norm_out"):
            s[p + ln + ".weight"] = (D_MODEL,)
            s[p + ln + ".bias"] = (D_MODEL,)
        for ff in ("feed_forward1", "feed_forward2"):
            s[p + ff + ".linear1.weight"] = (FF, D_MODEL)
            s[p + ff + ".linear1.bias"] = (FF,)
            s[p + ff + ".linear2.weight"] = (D_MODEL, FF)
            s[p + ff + ".linear2.bias"] = (D_MODEL,)
        for lin in ("linear_q", "linear_k", "linear_v", "linear_out"):
            s[p + "self_attn." + lin + ".weight"] = (D_MODEL, D_MODEL)
            s[p + "self_attn." + lin + ".bias"] = (D_MODEL,)
        s[p + "self_attn.linear_pos.weight"] = (D_MODEL, D_MODEL)
        s[p + "self_attn.pos_bias_u"] = (N_HEADS, D_K)
        s[p + "self_attn.pos_bias_v"] = (N_HEADS, D_K)
        s[p + "conv.pointwise_conv1.weight"] = (2 * D_MODEL, D_MODEL, 1)
        s[p + "conv.pointwise_conv1.bias"] = (2 * D_MODEL,)
        s[p + "conv.depthwise_conv.weight"] = (D_MODEL, 1, CONV_K)
        s[p + "conv.depthwise_conv.bias"] = (D_MODEL,)
        s[p + "conv.batch_norm.weight"] = (D_MODEL,)
        s[p + "conv.batch_norm.bias"] = (D_MODEL,)
        s[p + "conv.batch_norm.running_mean"] = (D_MODEL,)
        s[p + "conv.batch_norm.running_var"] = (D_MODEL,)
        s[p + "conv.pointwise_conv2.weight"] = (D_MODEL, D_MODEL, 1)
        s[p + "conv.pointwise_conv2.bias"] = (D_MODEL,)
    return s


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _noise(n: int, key: int) -> np.ndarray:
    """zero-mean, unit-variance-ish (Irwin-Hall of 4 bytes), exact in float32; the HIP library's
    own generator (csrc/qv_model.hip::init_random) is the same integer recipe."""
    with np.errstate(over="ignore"):
        h = _splitmix64(np.arange(n, dtype=np.uint64) + np.uint64(key))
    s = ((h & np.uint64(0xFF)) + ((h >> np.uint64(8)) & np.uint64(0xFF)) + ((h >> np.uint64(16)) & np.uint64(0xFF))
         + ((h >> np.uint64(24)) & np.uint64(0xFF))).astype(np.int64) - 510
    return s.astype(np.float32) * np.float32(1.0 / 147.8)


def init_rule(name: str, shape) -> tuple[float, float, bool]:
    """(offset, scale, abs) of the seeded init: value = offset + scale * (|noise| if abs else noise)."""
    n = int(np.prod(shape))
    if name.endswith("running_var"):
        return 1.0, 0.1, True
    if name.endswith("running_mean"):
        return 0.0, 0.1, False
    if ".norm_" in name or "batch_norm" in name:
        return (1.0, 0.1, False) if name.endswith("weight") else (0.0, 0.1, False)
    if name.endswith("bias") or "pos_bias" in name:
        return 0.0, 0.1, False
    fan_in = n // shape[0]
    return 0.0, float(np.float32(1.0) / np.sqrt(np.float32(fan_in))), False


def random_weights(seed: int = 20260630, n_layers: int = N_LAYERS) -> dict[str, torch.Tensor]:
    out = {}
    for name, shape in weight_shapes(n_layers).items():
        n = int(np.prod(shape))
        key = (_fnv1a(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
        z = _noise(n, key)
        off, sc, ab = init_rule(name, shape)
        if ab:
            z = np.abs(z)
        v = np.float32(off) + np.float32(sc) * z
        out[name] = torch.from_numpy(v.astype(np.float32).reshape(shape))
    return out


def structured_weights(seed: int = 20260630, rank: int = 16, mix: float = 0.15, head_gain: float = 6.0,
                       blank_bias: float = 3.0, n_layers: int = N_LAYERS) -> dict[str, torch.Tensor]:
    """Seeded weights with STRUCTURE, for the quantised-arithmetic noise-floor measurements (tools/ort_noise_floor.py,
    tools/ort_delta.py, tests/test_gpu_ort_mixed.py): i.i.d. N(0, 1/fan_in) matrices make every activation tensor
    unstructured and every posterior near-uniform -- the worst case for rounding-boundary flips.
    Here every weight matrix is a rank-`rank` product plus `mix` of the i.i.d. matrix (same Frobenius norm, so the
    activation scales of random_weights() are kept: trained layers are far from full rank), and the CTC head is
    `head_gain` times larger with `blank_bias` added to the blank logit (a trained CTC model emits peaked posteriors,
    mostly blank).  Still not a trained model -- the real file is absent -- but the two properties the objection names."""
    w = random_weights(seed, n_layers)
    rng = np.random.default_rng(seed ^ 0x5EED)
    out = {}
    for name, t in w.items():
        a = t.numpy()
        shape = a.shape
        # matrices: Linear [N,K], 1x1 / pointwise convolutions [N,K,1(,1)]; depthwise / 3x3 stencils and vectors stay
        is_matrix = name.endswith("weight") and a.ndim >= 2 and shape[0] >= 64 and int(np.prod(shape[1:])) >= 64
        if is_matrix:
            m = a.reshape(shape[0], -1).astype(np.float64)
            u = rng.standard_normal((m.shape[0], rank))
            v = rng.standard_normal((rank, m.shape[1]))
            lr = u @ v
            lr *= np.linalg.norm(m) / np.linalg.norm(lr)
            m2 = (1.0 - mix) * lr + mix * m
            m2 *= np.linalg.norm(m) / np.linalg.norm(m2)
            a = m2.reshape(shape).astype(np.float32)
        out[name] = torch.from_numpy(np.ascontiguousarray(a))
    head = "ctc_decoder.decoder_layers.0."
    out[head + "weight"] = out[head + "weight"] * np.float32(head_gain)
    b = out[head + "bias"].clone()
    b[VOCAB - 1] += np.float32(blank_bias)
    out[head + "bias"] = b
    return out


def damped_weights(seed: int = 20260630, branch: float = 0.25, head: float = 0.1, n_layers: int = N_LAYERS) -> dict[str, torch.Tensor]:
    """random_weights() with every residual branch's OUTPUT matrix (feed_forward*.linear2, self_attn.linear_out,
    conv.pointwise_conv2) scaled by `branch` and the CTC head by `head`: every quantiser of the ORT-mixed arithmetic still
    sees full-range activations, but a rounding-boundary flip inside a branch moves the residual stream -- and the log-probs --
    by correspondingly less.  The weight set on which the oracle's own reproducibility floor falls BELOW north_star's 1e-2, so
    that the device can be held to the absolute number there (tests/test_gpu_ort_mixed.py, tools/ort_floor_table.py)."""
    w = random_weights(seed, n_layers)
    out = {}
    for name, t in w.items():
        scale = 1.0
        if name.endswith(("feed_forward1.linear2.weight", "feed_forward2.linear2.weight", "self_attn.linear_out.weight",
                          "conv.pointwise_conv2.weight")):
            scale = branch
        elif name == "ctc_decoder.decoder_layers.0.weight":
            scale = head
        out[name] = (t * np.float32(scale)).contiguous() if scale != 1.0 else t
    return out


# ------------------------------------------------------------------- int4 weights -----
# The reference's "mixed" model file stores the Linear-layer MatMuls as 4-bit MatMulNBits
# ("MatMulNBitsQuantizer int4", experiments/c2c-direct-mixed/run.py:1-9; the script it names,
# scripts/quantize_mixed.py, is not part of /root/reference, so block size and symmetry are the
# onnxruntime defaults at best).  [EXT: onnxruntime's quantiser is not in /root/reference either;
# the rule below (block 128, scale = extreme value / -8, zero point 8) restates MLAS's symmetric
# Q4 blockwise quantiser and is UNPINNED like the rest of this file.  The device format carries a
# per-block zero point, so an asymmetric file maps onto it unchanged.]
# qv_pack_w4 (offline-tarteel_amd/csrc/qv_gemm.hip) performs the identical f32 arithmetic.
def quant_dequant_int4(w2d: np.ndarray, block: int = 128) -> np.ndarray:
    """[N][K] f32 -> the f32 matrix the W4A16 GEMM effectively multiplies by."""
    w2d = np.ascontiguousarray(w2d, dtype=np.float32)
    N, K = w2d.shape
    assert K % block == 0
    wb = w2d.reshape(N, K // block, block)
    idx = np.abs(wb).argmax(-1)                                  # first element of largest magnitude
    vmax = np.take_along_axis(wb, idx[..., None], -1)[..., 0]
    scale = (vmax / np.float32(-8.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        rs = np.where(scale != 0, np.float32(1.0) / scale, np.float32(0.0)).astype(np.float32)
    t = (wb * rs[..., None]).astype(np.float32) + np.float32(8.0)
    q = np.clip(np.floor(t + np.float32(0.5)), 0, 15).astype(np.float32)
    sh = scale.astype(np.float16).astype(np.float32)
    return ((q - np.float32(8.0)) * sh[..., None]).astype(np.float32).reshape(N, K)


def quant_dequant_int8(w2d: np.ndarray) -> np.ndarray:
    """[N][K] f32 -> the f32 matrix the W8A16 GEMM effectively multiplies by: per-row symmetric int8,
    scale = max|w| / 127 (1 for an all-zero row), q = clip(floor(w / scale + 0.5), -127, 127)
    (csrc/qv_gemm.hip::qv_pack_w8)."""
    w2d = np.ascontiguousarray(w2d, dtype=np.float32)
    amax = np.abs(w2d).max(axis=1, keepdims=True).astype(np.float32)
    scale = np.where(amax > 0, amax / np.float32(127.0), np.float32(1.0)).astype(np.float32)
    t = (w2d / scale).astype(np.float32)
    q = np.clip(np.floor(t + np.float32(0.5)), -127, 127).astype(np.float32)
    return (q * scale).astype(np.float32)


INT8_CONV_SUFFIXES = ("conv.pointwise_conv1.weight", "conv.pointwise_conv2.weight")

INT4_LINEAR_SUFFIXES = (
    "feed_forward1.linear1.weight", "feed_forward1.linear2.weight",
    "feed_forward2.linear1.weight", "feed_forward2.linear2.weight",
    "self_attn.linear_q.weight", "self_attn.linear_k.weight", "self_attn.linear_v.weight",
    "self_attn.linear_out.weight", "self_attn.linear_pos.weight",
)


def quantize_linear_weights(w: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
    """weights as the QV_PREC_MIXED_INT4_INT8 engine sees them: Linear layers of the 17 Conformer
    blocks through int4 quantise -> dequantise, the two pointwise convolutions of each conv module
    through per-channel int8, everything else untouched."""
    out = dict(w)
    for name, t in w.items():
        if name.startswith("encoder.layers.") and name.endswith(INT4_LINEAR_SUFFIXES):
            out[name] = torch.from_numpy(quant_dequant_int4(t.numpy()))
        elif name.startswith("encoder.layers.") and name.endswith(INT8_CONV_SUFFIXES):
            a = t.numpy()
            out[name] = torch.from_numpy(quant_dequant_int8(a.reshape(a.shape[0], -1)).reshape(a.shape))
    return out


# ------------------------------------------------------------------- onnxruntime semantics ----
# What the reference's model file actually computes (experiments/c2c-direct-mixed/run.py:1-9: "step 1
# MatMulNBitsQuantizer int4 on the MatMul weights, step 2 quantize_dynamic QInt8 on what remains", run by
# onnxruntime's CPU provider with fp32 activations).  [EXT / UNPINNED like everything in this file: the model
# file and onnxruntime are absent; the rules below restate onnxruntime's published operator definitions.]
#   com.microsoft.MatMulNBits (bits 4, block 128, symmetric, zero point 8): y = x @ ((q - 8) * scale)^T with
#       FLOAT32 scales and fp32 accumulation -- the device path keeps the scale in fp16 (5e-4 relative).
#   DynamicQuantizeLinear + ConvInteger (what quantize_dynamic makes of a Conv): the ACTIVATION tensor of the
#       call is quantised to uint8 with one scale / zero point (range widened to include 0, round half to
#       even, saturate), the weight is int8 with one per-tensor symmetric scale, the convolution accumulates
#       in int32 and the result is scaled back by scale_x * scale_w and biased in fp32.
# `OrtMixed` carries the choice of which tensors get which treatment; forward(..., ort=OrtMixed()) routes
# every Linear / Conv of the model through it.  This is the yardstick for "how far is the HIP mixed path
# (W4A16 / W8A16: weights dequantised, fp16 activations, no activation quantisation) from the arithmetic the
# reference ran" -- tests/test_gpu_forward.py reports the max |delta log-prob|.
ORT_INT4_SUFFIXES = INT4_LINEAR_SUFFIXES + ("encoder.pre_encode.out.weight",)


def quant_dequant_int4_f32scale(w2d: np.ndarray, block: int = 128) -> np.ndarray:
    """as quant_dequant_int4 but with the block scale kept in float32 (MatMulNBits on an fp32 model)."""
    w2d = np.ascontiguousarray(w2d, dtype=np.float32)
    N, K = w2d.shape
    assert K % block == 0
    wb = w2d.reshape(N, K // block, block)
    idx = np.abs(wb).argmax(-1)
    vmax = np.take_along_axis(wb, idx[..., None], -1)[..., 0]
    scale = (vmax / np.float32(-8.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        rs = np.where(scale != 0, np.float32(1.0) / scale, np.float32(0.0)).astype(np.float32)
    q = np.clip(np.floor((wb * rs[..., None]).astype(np.float32) + np.float32(8.5)), 0, 15).astype(np.float32)
    return ((q - np.float32(8.0)) * scale[..., None]).astype(np.float32).reshape(N, K)


def dynamic_quantize_linear(x: torch.Tensor):
    """onnx DynamicQuantizeLinear of one tensor: (x_q as integer-valued float64, scale, zero_point), evaluated in
    FLOAT32 like onnxruntime's CPU kernel does (range widened to include 0; scale = (max - min) / 255; zero point =
    round-half-even(0 - min / scale) saturated to [0, 255]; x_q = saturate(round-half-even(x / scale) + zero_point)).
    [EXT] -- the HIP path (csrc/qv_ort.h::dql_param / quant_u8) performs the identical float32 operations."""
    xmin = np.float32(min(0.0, float(x.min())))
    xmax = np.float32(max(0.0, float(x.max())))
    scale = np.float32((xmax - xmin) / np.float32(255.0))                             # float32 subtraction and division
    if scale == 0:
        return torch.zeros_like(x, dtype=torch.float64), np.float32(1.0), 0.0
    zp = float(np.clip(np.rint(np.float32(-xmin) / scale), 0, 255))                   # np.rint: half to even
    xq = torch.clamp(torch.round(x / float(scale)) + zp, 0, 255).to(torch.float64)     # true f32 division, half to even
    return xq, scale, zp


def quantize_weight_int8(wt: torch.Tensor):
    """quantize_dynamic(weight_type=QInt8) on a Conv weight: ONE symmetric scale per tensor, scale = max|w| / 127,
    q = saturate(round-half-even(w / scale)) to [-127, 127].  Returns (q as integer-valued float64, scale f32).
    csrc/qv_model.hip::quant_w8_tensor performs the identical arithmetic."""
    amax = float(wt.abs().max())
    sw = np.float32(amax / 127.0) if amax > 0 else np.float32(1.0)
    return torch.clamp(torch.round(wt / float(sw)), -127, 127).to(torch.float64), sw


class OrtMixed:
    """Routing of the model's Linear / Conv calls through the onnxruntime arithmetic above."""

    def __init__(self, int4_linears: bool = True, int8_convs="all", conv_scales: dict | None = None):
        """int4_linears=False + conv_scales={weight name: float32 scale}: the weights handed to forward() are the
        DEQUANTISED contents of an already quantised model file (MatMulNBits values used as they are; each Conv weight
        put back on its integers with the file's own scale) -- what tools/convert_weights.py --onnx produces."""
        self.int4_linears = int4_linears
        self.int8_convs = int8_convs          # "all", or a tuple of weight-name suffixes, or ()
        self.conv_scales = dict(conv_scales or {})
        self._w4 = {}
        self._w8 = {}

    def _is_conv8(self, name: str) -> bool:
        return self.int8_convs == "all" or (bool(self.int8_convs) and name.endswith(tuple(self.int8_convs)))

    def linear(self, w, name: str, x: torch.Tensor, bias_name: str | None):
        wt = w[name]
        if self.int4_linears and name.endswith(ORT_INT4_SUFFIXES):
            if name not in self._w4:
                self._w4[name] = torch.from_numpy(quant_dequant_int4_f32scale(wt.numpy()))
            wt = self._w4[name]
        return F.linear(x, wt, w[bias_name] if bias_name else None)

    def conv(self, w, name: str, x: torch.Tensor, bias_name: str | None, fn, **kw):
        """fn = F.conv1d / F.conv2d.  Quantisation is per CALL of the reference, i.e. per utterance (it
        feeds batch 1): every batch item gets its own activation scale.  forward(..., ort=...) therefore runs
        one utterance at a time, unpadded, so that a call's range never sees padding frames."""
        wt = w[name]
        bias = w[bias_name] if bias_name else None
        if not self._is_conv8(name):
            return fn(x, wt, bias, **kw)
        if name not in self._w8:
            if name in self.conv_scales:
                sw = np.float32(self.conv_scales[name])
                self._w8[name] = (torch.clamp(torch.round(wt / float(sw)), -127, 127).to(torch.float64), sw)
            else:
                self._w8[name] = quantize_weight_int8(wt)
        wq, sw = self._w8[name]
        outs = []
        for b in range(x.shape[0]):
            xq, sx, zp = dynamic_quantize_linear(x[b: b + 1])
            acc = fn(xq - zp, wq, None, **kw)                       # exact: integer-valued float64
            # int32 accumulator -> float32, times the float32 product of the two scales (one rounding), plus bias
            y = (acc * float(np.float32(sx) * np.float32(sw))).to(torch.float32)
            outs.append(y + bias.view(1, -1, *([1] * (y.dim() - 2))) if bias is not None else y)
        return torch.cat(outs, 0)


class _Plain:
    """fp32 routing (the default): plain F.linear / conv."""

    def linear(self, w, name, x, bias_name):
        return F.linear(x, w[name], w[bias_name] if bias_name else None)

    def conv(self, w, name, x, bias_name, fn, **kw):
        return fn(x, w[name], w[bias_name] if bias_name else None, **kw)


# ------------------------------------------------------------------- front-end --------
def mel_filterbank() -> np.ndarray:
    """librosa.filters.mel(sr=16000, n_fft=512, n_mels=80, fmin=0, fmax=8000, htk=False,
    norm='slaney') restated (Slaney auditory-toolbox scale)."""
    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        f_sp = 200.0 / 3
        min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, SR / 2, 1 + N_FFT // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(SR / 2), N_MELS + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((N_MELS, 1 + N_FFT // 2))
    for i in range(N_MELS):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: N_MELS + 2] - mel_f[:N_MELS])
    return (w * enorm[:, None]).astype(np.float32)


def hann_window() -> torch.Tensor:
    return torch.hann_window(WIN, periodic=False, dtype=torch.float32)


def mel_frames(n_samples: int) -> int:
    return n_samples // HOP + 1


def sub_len(t: int) -> int:
    for _ in range(3):
        t = (t + 2 - 3) // 2 + 1
    return t


def frontend(audio: torch.Tensor, lengths: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """audio [B,N] float32 (zero padded), lengths [B] -> features [B, 80, Tm] and Tm lengths."""
    B, N = audio.shape
    tm = torch.div(lengths, HOP, rounding_mode="floor") + 1
    x = torch.cat([audio[:, :1], audio[:, 1:] - PREEMPH * audio[:, :-1]], dim=1)
    # samples past each utterance's length must behave like the unpadded case: the reference feeds
    # every file alone, and torch.stft's reflect padding mirrors around ITS end.  Process per item.
    fb = torch.from_numpy(mel_filterbank())
    win = hann_window()
    Tm = int(tm.max())
    out = torch.zeros(B, N_MELS, Tm)
    for b in range(B):
        n = int(lengths[b])
        st = torch.stft(x[b, :n], N_FFT, hop_length=HOP, win_length=WIN, window=win, center=True,
                        pad_mode="reflect", return_complex=True)
        p = st.real ** 2 + st.imag ** 2
        p = torch.sqrt(p) ** 2  # NeMo: magnitude then .pow(2)
        m = torch.log(fb @ p + LOG_GUARD)
        t = int(tm[b])
        m = m[:, :t]
        mean = m.mean(dim=1, keepdim=True)
        std = torch.sqrt(((m - mean) ** 2).sum(dim=1, keepdim=True) / (t - 1)) + 1e-5
        out[b, :, :t] = (m - mean) / std
    return out, tm


# ------------------------------------------------------------------- encoder ----------
def rel_pos_emb(T: int) -> torch.Tensor:
    """RelPositionalEncoding: positions T-1 .. -(T-1) -> [2T-1, d_model]."""
    pos = torch.arange(T - 1, -T, -1, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, D_MODEL, 2, dtype=torch.float32) * -(math.log(10000.0) / D_MODEL))
    pe = torch.zeros(2 * T - 1, D_MODEL)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def rel_shift(x: torch.Tensor) -> torch.Tensor:
    b, h, qlen, pos_len = x.size()
    x = F.pad(x, pad=(1, 0))
    x = x.view(b, h, -1, qlen)
    return x[:, :, 1:].view(b, h, qlen, pos_len)


def subsampling(w, feats: torch.Tensor, tm: torch.Tensor, ops=None, taps: dict | None = None):
    """feats [B,80,Tm] -> [B,T,512]; activations past each stage's valid length are zeroed so a
    padded batch equals the per-utterance (unpadded) result.  taps (optional) receives the
    channels-last stage outputs 'c1' (conv.2), 'c1p' (ReLU conv.3), 'c2' (conv.5), 'c2p' (ReLU conv.6): [B,T,F,256]."""
    x = feats.transpose(1, 2).unsqueeze(1)  # [B,1,Tm,80]
    lens = tm.clone()

    def mask_t(x, lens):
        T = x.shape[2]
        m = (torch.arange(T)[None, :] < lens[:, None]).to(x.dtype)
        return x * m[:, None, :, None]

    ops = ops or _Plain()
    pe = "encoder.pre_encode."
    x = mask_t(x, lens)
    x = F.relu(ops.conv(w, pe + "conv.0.weight", x, pe + "conv.0.bias", F.conv2d, stride=2, padding=1))
    lens = (lens + 2 - 3) // 2 + 1
    x = mask_t(x, lens)
    for k, (dw, pw) in enumerate(((2, 3), (5, 6))):
        x = ops.conv(w, f"{pe}conv.{dw}.weight", x, f"{pe}conv.{dw}.bias", F.conv2d, stride=2, padding=1, groups=SUB_CH)
        if taps is not None:
            taps[f"c{k + 1}"] = x.permute(0, 2, 3, 1).contiguous()
        x = F.relu(ops.conv(w, f"{pe}conv.{pw}.weight", x, f"{pe}conv.{pw}.bias", F.conv2d))
        lens = (lens + 2 - 3) // 2 + 1
        x = mask_t(x, lens)
        if taps is not None:
            taps[f"c{k + 1}p"] = x.permute(0, 2, 3, 1).contiguous()
    b, c, t, f = x.shape
    x = x.transpose(1, 2).reshape(b, t, c * f)
    x = ops.linear(w, pe + "out.weight", x, pe + "out.bias")
    return x, lens


def conformer_layer(w, p: str, x: torch.Tensor, pos_emb: torch.Tensor, pad: torch.Tensor, ops=None, taps: dict | None = None,
                    tag: str = ""):
    """x [B,T,512]; pad [B,T] True where padded.  taps (optional) receives the conv module's stage tensors
    'lnc<tag>' (norm_conv output), 'glu<tag>' (GLU output), 'dw<tag>' (depthwise conv + BatchNorm + Swish): [B,T,512]."""
    B, T, _ = x.shape
    ops = ops or _Plain()

    def ln(name, t):
        return F.layer_norm(t, (D_MODEL,), w[p + name + ".weight"], w[p + name + ".bias"], 1e-5)

    def ffn(name, t):
        t = ops.linear(w, p + name + ".linear1.weight", t, p + name + ".linear1.bias")
        t = t * torch.sigmoid(t)
        return ops.linear(w, p + name + ".linear2.weight", t, p + name + ".linear2.bias")

    r = x
    r = r + 0.5 * ffn("feed_forward1", ln("norm_feed_forward1", r))
    # --- rel-pos MHSA
    y = ln("norm_self_att", r)
    a = p + "self_attn."
    q = ops.linear(w, a + "linear_q.weight", y, a + "linear_q.bias").view(B, T, N_HEADS, D_K)
    k = ops.linear(w, a + "linear_k.weight", y, a + "linear_k.bias").view(B, T, N_HEADS, D_K).transpose(1, 2)
    v = ops.linear(w, a + "linear_v.weight", y, a + "linear_v.bias").view(B, T, N_HEADS, D_K).transpose(1, 2)
    pp = ops.linear(w, a + "linear_pos.weight", pos_emb, None).view(1, -1, N_HEADS, D_K).transpose(1, 2)
    qu = (q + w[a + "pos_bias_u"]).transpose(1, 2)
    qv = (q + w[a + "pos_bias_v"]).transpose(1, 2)
    bd = rel_shift(torch.matmul(qv, pp.transpose(-2, -1)))
    ac = torch.matmul(qu, k.transpose(-2, -1))
    bd = bd[:, :, :, : ac.size(-1)]
    scores = (ac + bd) / math.sqrt(D_K)
    valid = ~pad
    att_mask = ~(valid[:, None, :] & valid[:, :, None])  # [B,T,T] True = masked
    scores = scores.masked_fill(att_mask[:, None], -10000.0)
    attn = torch.softmax(scores, dim=-1).masked_fill(att_mask[:, None], 0.0)
    ctx = torch.matmul(attn, v).transpose(1, 2).reshape(B, T, D_MODEL)
    r = r + ops.linear(w, a + "linear_out.weight", ctx, a + "linear_out.bias")
    # --- conv module
    y = ln("norm_conv", r).transpose(1, 2)
    if taps is not None:
        taps["lnc" + tag] = y.transpose(1, 2).contiguous()
    c = p + "conv."
    y = ops.conv(w, c + "pointwise_conv1.weight", y, c + "pointwise_conv1.bias", F.conv1d)
    y = F.glu(y, dim=1)
    y = y.masked_fill(pad[:, None, :], 0.0)
    if taps is not None:
        taps["glu" + tag] = y.transpose(1, 2).contiguous()
    y = ops.conv(w, c + "depthwise_conv.weight", y, c + "depthwise_conv.bias", F.conv1d, padding=(CONV_K - 1) // 2, groups=D_MODEL)
    y = F.batch_norm(y, w[c + "batch_norm.running_mean"], w[c + "batch_norm.running_var"], w[c + "batch_norm.weight"],
                     w[c + "batch_norm.bias"], False, 0.0, 1e-5)
    y = y * torch.sigmoid(y)
    if taps is not None:
        taps["dw" + tag] = y.transpose(1, 2).contiguous()
    y = ops.conv(w, c + "pointwise_conv2.weight", y, c + "pointwise_conv2.bias", F.conv1d).transpose(1, 2)
    r = r + y
    r = r + 0.5 * ffn("feed_forward2", ln("norm_feed_forward2", r))
    return ln("norm_out", r)


@torch.no_grad()
def forward(w, audio: torch.Tensor, lengths, n_layers: int = N_LAYERS, taps: dict | None = None, ort: OrtMixed | None = None):
    """audio [B,N] float32, lengths -> (log_probs [B,T,1025], T lengths).  taps (optional dict)
    receives intermediate activations: 'mel' [B,Tm,80], 'sub' [B,T,512], 'layer{i}' [B,T,512] (+ the stage tensors
    of subsampling() / conformer_layer(): 'c1', 'c1p', 'c2', 'c2p', 'lnc{i}', 'glu{i}', 'dw{i}').
    ort: route every Linear / Conv through the onnxruntime int4 / dynamic-int8 arithmetic (OrtMixed).  The
    reference feeds onnxruntime ONE unpadded utterance per call (experiments/c2c-direct-mixed/run.py:59-63) and the
    activation ranges of DynamicQuantizeLinear are per call, so with `ort` every utterance runs alone, trimmed to its
    length; results (and taps, zero padded) are stacked afterwards."""
    lengths = torch.as_tensor(lengths, dtype=torch.int64)
    if ort is not None and (audio.shape[0] > 1 or int(lengths[0]) != audio.shape[1]):
        outs, lens, per = [], [], []
        for b in range(audio.shape[0]):
            n = int(lengths[b])
            tb = {} if taps is not None else None
            lp, tl = forward(w, audio[b: b + 1, :n].contiguous(), [n], n_layers, tb, ort)
            outs.append(lp[0])
            lens.append(int(tl[0]))
            per.append(tb)
        T = max(lens)
        lp = torch.zeros(len(outs), T, VOCAB)
        for b, o in enumerate(outs):
            lp[b, : o.shape[0]] = o
        if taps is not None:
            for k in per[0]:
                shp = [max(t[k].shape[1] for t in per)] + list(per[0][k].shape[2:])
                z = torch.zeros(len(per), *shp)
                for b, t in enumerate(per):
                    z[b, : t[k].shape[1]] = t[k][0]
                taps[k] = z
        return lp, torch.tensor(lens, dtype=torch.int64)
    feats, tm = frontend(audio, lengths)
    if taps is not None:
        taps["mel"] = feats.transpose(1, 2).contiguous()
    x, lens = subsampling(w, feats, tm, ort, taps)
    if taps is not None:
        taps["sub"] = x.clone()
    B, T, _ = x.shape
    x = x * math.sqrt(D_MODEL)
    pos_emb = rel_pos_emb(T).unsqueeze(0)
    pad = torch.arange(T)[None, :] >= lens[:, None]
    for i in range(n_layers):
        x = conformer_layer(w, f"encoder.layers.{i}.", x, pos_emb, pad, ort, taps, str(i))
        if taps is not None:
            taps[f"layer{i}"] = x.clone()
    head = "ctc_decoder.decoder_layers.0."
    if ort is not None:   # ConvASRDecoder is a 1x1 Conv1d in the exported graph
        logits = ort.conv(w, head + "weight", x.transpose(1, 2), head + "bias", F.conv1d).transpose(1, 2)
    else:
        logits = F.linear(x, w[head + "weight"].squeeze(-1), w[head + "bias"])
    return torch.log_softmax(logits, dim=-1), lens
