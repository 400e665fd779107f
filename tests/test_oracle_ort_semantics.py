"""oracle/fastconformer_ref.py::OrtMixed -- the onnxruntime arithmetic of the reference's model file
(int4 MatMulNBits + dynamic-int8 ConvInteger, experiments/c2c-direct-mixed/run.py:1-9) restated for the
CPU oracle.  [EXT: onnxruntime is absent; these tests pin the restatement to the operator definitions
(onnx DynamicQuantizeLinear / ConvInteger formulas written out in numpy integers), not to onnxruntime.]"""

import numpy as np
import torch
import torch.nn.functional as F

from oracle import fastconformer_ref as R


def test_dynamic_quantize_linear_is_the_onnx_formula():
    rng = np.random.default_rng(0)
    for shape, lo, hi in (((1, 8, 50), -3.0, 2.0), ((1, 4, 7), 0.5, 4.0), ((1, 3, 9), -2.0, -0.25), ((1, 2, 5), 0.0, 0.0)):
        x = rng.uniform(lo, hi, size=shape).astype(np.float32)
        xq, scale, zp = R.dynamic_quantize_linear(torch.from_numpy(x))
        xmin, xmax = min(0.0, float(x.min())), max(0.0, float(x.max()))
        if xmax == xmin:
            assert float(xq.abs().max()) == 0.0
            continue
        s = np.float32((xmax - xmin) / 255.0)
        z = np.clip(np.round(np.float32(0.0 - xmin) / s), 0, 255)
        want = np.clip(np.round(x / s) + z, 0, 255)
        assert scale == s and zp == z
        assert np.array_equal(xq.numpy(), want.astype(np.float64))
        assert xq.min() >= 0 and xq.max() <= 255


def test_conv_integer_path_is_exact_integer_arithmetic():
    rng = np.random.default_rng(1)
    w = {"c.weight": torch.from_numpy(rng.normal(size=(6, 4, 3)).astype(np.float32)),
         "c.bias": torch.from_numpy(rng.normal(size=(6,)).astype(np.float32))}
    x = torch.from_numpy(rng.normal(size=(2, 4, 20)).astype(np.float32) * 3)
    ops = R.OrtMixed()
    y = ops.conv(w, "c.weight", x, "c.bias", F.conv1d, padding=1)
    sw = np.float32(float(w["c.weight"].abs().max()) / 127.0)
    wq = np.clip(np.round(w["c.weight"].numpy() / sw), -127, 127).astype(np.int64)
    for b in range(2):                                  # one activation scale per utterance (the reference feeds batch 1)
        xq, sx, zp = R.dynamic_quantize_linear(x[b: b + 1])
        xi = np.pad(xq.numpy()[0].astype(np.int64) - int(zp), ((0, 0), (1, 1)))
        acc = np.zeros((6, 20), np.int64)
        for o in range(6):
            for t in range(20):
                acc[o, t] = int((wq[o] * xi[:, t: t + 3]).sum())
        want = (acc.astype(np.float64) * float(np.float32(sx) * sw)).astype(np.float32) + w["c.bias"].numpy()[:, None]
        assert np.array_equal(y[b].numpy(), want)


def test_int4_f32_scale_differs_from_device_rule_only_by_the_scale_rounding():
    rng = np.random.default_rng(2)
    w = rng.normal(size=(64, 256)).astype(np.float32) * 0.04
    a, b = R.quant_dequant_int4_f32scale(w), R.quant_dequant_int4(w)
    # same codes; the device keeps half(scale): at most 2^-11 relative per weight
    nz = a != 0
    assert np.abs(b[nz] / a[nz] - 1).max() <= 2.0 ** -11 + 1e-7
    assert np.array_equal(a == 0, b == 0)


def test_ort_forward_runs_and_is_closer_to_the_device_weights_than_to_fp32():
    from synth import synth_audio

    w = R.random_weights(7, n_layers=2)
    a = torch.from_numpy(synth_audio(1, 16000))
    lp_ort, T = R.forward(w, a, [16000], n_layers=2, ort=R.OrtMixed())
    lp32, _ = R.forward(w, a, [16000], n_layers=2)
    lpq, _ = R.forward(R.quantize_linear_weights(w), a, [16000], n_layers=2)
    assert lp_ort.shape == lp32.shape and torch.isfinite(lp_ort).all()
    d32, dq = float((lp_ort - lp32).abs().max()), float((lp_ort - lpq).abs().max())
    assert 0 < dq < d32


def test_ort_forward_is_per_utterance_like_the_reference_feeds_it():
    """The reference hands onnxruntime ONE unpadded utterance per call and DynamicQuantizeLinear takes its range from
    the call's tensor: under `ort` a ragged batch must equal each utterance run alone, bit for bit, taps included."""
    from synth import synth_audio

    w = R.random_weights(7, n_layers=2)
    lens = [16000, 11000]
    a = torch.from_numpy(synth_audio(2, 16000))
    a[1, lens[1]:] = 0
    taps = {}
    lp, T = R.forward(w, a, lens, n_layers=2, taps=taps, ort=R.OrtMixed())
    for b, n in enumerate(lens):
        t1 = {}
        one, To = R.forward(w, a[b: b + 1, :n].contiguous(), [n], n_layers=2, taps=t1, ort=R.OrtMixed())
        t = int(To[0])
        assert int(T[b]) == t and torch.equal(one[0, :t], lp[b, :t])
        for k in ("mel", "c1", "c1p", "c2", "c2p", "sub", "lnc1", "glu1", "dw1", "layer1"):
            assert torch.equal(taps[k][b, : t1[k].shape[1]], t1[k][0]), k
    assert float(taps["glu0"][1, int(T[1]):].abs().max()) == 0.0      # stacked taps are zero padded


def test_batch_norm_eval_formula_the_device_uses_is_torchs():
    """csrc/qv_ort.hip::k_dwconv1d_ort applies BatchNorm as fma(y, alpha, beta) with alpha = gamma * (1 / sqrt(var + eps)),
    beta = fma(-mean, alpha, bias) (prepared in csrc/qv_model.hip): torch's eval-mode batch_norm on the CPU, bit for bit."""
    rng = np.random.default_rng(3)
    C, T = 512, 97
    y = rng.standard_normal((1, C, T)).astype(np.float32) * 3
    mean, var = rng.standard_normal(C).astype(np.float32) * 0.1, np.abs(rng.standard_normal(C).astype(np.float32)) * 0.1 + 1
    g, b = 1 + rng.standard_normal(C).astype(np.float32) * 0.1, rng.standard_normal(C).astype(np.float32) * 0.1
    ref = F.batch_norm(torch.from_numpy(y), torch.from_numpy(mean), torch.from_numpy(var), torch.from_numpy(g), torch.from_numpy(b),
                       False, 0.0, 1e-5).numpy()
    alpha = (g * (np.float32(1) / np.sqrt(var + np.float32(1e-5)).astype(np.float32))).astype(np.float32)
    beta = (b.astype(np.float64) - mean.astype(np.float64) * alpha.astype(np.float64)).astype(np.float32)          # fma(-mean, alpha, b)
    out = (y.astype(np.float64) * alpha[None, :, None].astype(np.float64) + beta[None, :, None].astype(np.float64)).astype(np.float32)
    assert np.array_equal(out, ref)


def test_weight_quantiser_is_one_symmetric_scale_per_tensor():
    rng = np.random.default_rng(4)
    w = torch.from_numpy(rng.normal(size=(8, 3, 9)).astype(np.float32))
    q, s = R.quantize_weight_int8(w)
    assert s == np.float32(float(w.abs().max()) / 127.0)
    assert float(q.abs().max()) == 127.0 and torch.equal(q, torch.round(q))
    assert float((q * float(s) - w).abs().max()) <= float(s) / 2 + 1e-7
