"""QV_PREC_ORT_MIXED on the GPU: the arithmetic onnxruntime runs on the reference's model file
(experiments/c2c-direct-mixed/run.py:1-9 -- MatMulNBits int4 + DynamicQuantizeLinear / ConvInteger on every Conv)
against its CPU restatement oracle/fastconformer_ref.py::OrtMixed, through the C ABI.

Two kinds of checks:
  * STAGE checks: the float32 tensor in front of a quantiser is read back from the device (qv_debug_forward_tap 3..9),
    the oracle runs that one Conv on it, and the device's own output tensor must equal the oracle's.  The integer part
    (quantised activations, int8 weights, int32 accumulation, float32 rescale + bias) has no tolerance to hide in: one
    wrong LSB is 4e-3 of the tensor's range; the bounds below are 1e-5 of it (Swish / sigmoid differ by an ulp or two
    between expf implementations, everything else is expected to be bit-identical and the match rate is printed).
  * END-TO-END: log-probs against the oracle.  The oracle differs from ITSELF by ~0.05 max / ~0.012 rms when only the
    float32 summation order changes (1 vs 16 threads; profiles/r03_a_ort_noise_floor.json), because every
    DynamicQuantizeLinear is a rounding discontinuity; the bound here is that floor with head room, and the f16-weight
    path (QV_PREC_MIXED_INT4_INT8) must be several times further away.
"""

import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from synth import synth_audio

pytestmark = pytest.mark.gpu

SEED = 7
LENS = [48000, 30000, 17777]


@pytest.fixture(scope="module")
def setup():
    os.environ["QVERSE_DEBUG_TAPS"] = "1"
    from offline_tarteel_amd.engine import Engine
    from oracle import fastconformer_ref as R

    torch.set_num_threads(min(16, torch.get_num_threads()))
    audio = torch.from_numpy(synth_audio(3, 48000))
    for b, n in enumerate(LENS):
        audio[b, n:] = 0
    w = R.random_weights(SEED)
    eng = Engine(device=0, with_model=True, seed=SEED, precision=2, max_batch=4, max_samples=80000)
    lp, t = eng.forward(audio.cuda().contiguous(), LENS)
    torch.cuda.synchronize()
    tm = [n // 160 + 1 for n in LENS]
    sl = lambda x: (x + 2 - 3) // 2 + 1  # noqa: E731
    l1 = [sl(x) for x in tm]
    l2 = [sl(x) for x in l1]
    l3 = [sl(x) for x in l2]
    assert l3 == t
    yield dict(eng=eng, audio=audio, lp=lp, t=t, w=w, R=R, tm=tm, l1=l1, l2=l2, l3=l3, ort=R.OrtMixed())
    eng.close()
    os.environ.pop("QVERSE_DEBUG_TAPS", None)


def _report(name, got, want, rel):
    got, want = got.float(), want.float()
    scale = float(want.abs().max())
    d = float((got - want).abs().max())
    same = float((got == want).float().mean())
    print(f"[ort-stage] {name}: max|d| {d:.3e} of range {scale:.3e}, bit-identical {same:.6f}")
    assert d <= rel * scale, (name, d, scale, same)


def test_frontend_integer_stages(setup):
    """normalised mel -> conv.0 + ReLU -> conv.2 | -> conv.3 + ReLU | -> conv.5 | -> conv.6 + ReLU, each stage on the
    DEVICE's own input tensor."""
    eng, w, ort = setup["eng"], setup["w"], setup["ort"]
    tm, l2, l3 = setup["tm"], setup["l2"], setup["l3"]
    B = len(LENS)
    pe = "encoder.pre_encode."
    mel = eng.forward_tap(0, 0, (B, max(tm), 80)).cpu()
    c1 = eng.forward_tap(6, 0, (B, max(l2), 20, 256)).cpu()
    c1p = eng.forward_tap(7, 0, (B, max(l2), 20, 256)).cpu()
    c2 = eng.forward_tap(8, 0, (B, max(l3), 10, 256)).cpu()
    c2p = eng.forward_tap(9, 0, (B, max(l3), 10, 256)).cpu()
    for b in range(B):
        x = mel[b, : tm[b]].unsqueeze(0).unsqueeze(0)                                   # [1,1,Tm,80]
        y = F.relu(ort.conv(w, pe + "conv.0.weight", x, pe + "conv.0.bias", F.conv2d, stride=2, padding=1))
        y = ort.conv(w, pe + "conv.2.weight", y, pe + "conv.2.bias", F.conv2d, stride=2, padding=1, groups=256)
        _report(f"conv.0+conv.2[{b}]", c1[b, : l2[b]], y[0].permute(1, 2, 0), 1e-6)
        x = c1[b, : l2[b]].permute(2, 0, 1).unsqueeze(0).contiguous()                   # [1,256,T2,20]
        y = F.relu(ort.conv(w, pe + "conv.3.weight", x, pe + "conv.3.bias", F.conv2d))
        _report(f"conv.3[{b}]", c1p[b, : l2[b]], y[0].permute(1, 2, 0), 1e-6)
        x = c1p[b, : l2[b]].permute(2, 0, 1).unsqueeze(0).contiguous()
        y = ort.conv(w, pe + "conv.5.weight", x, pe + "conv.5.bias", F.conv2d, stride=2, padding=1, groups=256)
        _report(f"conv.5[{b}]", c2[b, : l3[b]], y[0].permute(1, 2, 0), 1e-6)
        x = c2[b, : l3[b]].permute(2, 0, 1).unsqueeze(0).contiguous()
        y = F.relu(ort.conv(w, pe + "conv.6.weight", x, pe + "conv.6.bias", F.conv2d))
        _report(f"conv.6[{b}]", c2p[b, : l3[b]], y[0].permute(1, 2, 0).half().float(), 1e-6)   # stored as f16 (feeds the int4 Linear)


@pytest.mark.parametrize("layer", [0, 8, 16])
def test_conv_module_stages(setup, layer):
    """norm_conv output -> pointwise_conv1 + GLU | -> depthwise_conv + BatchNorm + Swish, on the device's own inputs."""
    eng, w, ort, T = setup["eng"], setup["w"], setup["ort"], setup["t"]
    B = len(LENS)
    c = f"encoder.layers.{layer}.conv."
    lnc = eng.forward_tap(3, layer, (B, max(T), 512)).cpu()
    glu = eng.forward_tap(4, layer, (B, max(T), 512)).cpu()
    dw = eng.forward_tap(5, layer, (B, max(T), 512)).cpu()
    for b in range(B):
        x = lnc[b, : T[b]].t().unsqueeze(0).contiguous()                                # [1,512,T]
        y = F.glu(ort.conv(w, c + "pointwise_conv1.weight", x, c + "pointwise_conv1.bias", F.conv1d), dim=1)
        _report(f"L{layer} pw1+GLU[{b}]", glu[b, : T[b]], y[0].t(), 1e-5)
        x = glu[b, : T[b]].t().unsqueeze(0).contiguous()
        y = ort.conv(w, c + "depthwise_conv.weight", x, c + "depthwise_conv.bias", F.conv1d, padding=4, groups=512)
        y = F.batch_norm(y, w[c + "batch_norm.running_mean"], w[c + "batch_norm.running_var"], w[c + "batch_norm.weight"],
                         w[c + "batch_norm.bias"], False, 0.0, 1e-5)
        y = y * torch.sigmoid(y)
        _report(f"L{layer} dw+BN+Swish[{b}]", dw[b, : T[b]], y[0].t(), 1e-5)


def test_whole_layer_and_head_on_device_inputs(setup):
    """one whole Conformer layer (covers pointwise_conv2's residual epilogue) and the CTC head, each from the device's own
    input: the Linear layers run f16-operand GEMMs here, so the bound is the f16 path's, with room for an LSB flip."""
    eng, w, ort, R, T = setup["eng"], setup["w"], setup["ort"], setup["R"], setup["t"]
    B = len(LENS)
    x0 = eng.forward_tap(1, 0, (B, max(T), 512)).cpu()
    x1 = eng.forward_tap(2, 0, (B, max(T), 512)).cpu()
    xe = eng.forward_tap(2, 16, (B, max(T), 512)).cpu()
    head = "ctc_decoder.decoder_layers.0."
    for b in range(B):
        t = T[b]
        pos = R.rel_pos_emb(t).unsqueeze(0)
        pad = torch.zeros(1, t, dtype=torch.bool)
        y = R.conformer_layer(w, "encoder.layers.0.", x0[b: b + 1, :t], pos, pad, ort)
        d = float((y[0] - x1[b, :t]).abs().max())
        print(f"[ort-stage] layer0[{b}]: max|d| {d:.3e} (values ~ +-3)")
        assert d <= 5e-2, d
        lg = ort.conv(w, head + "weight", xe[b: b + 1, :t].transpose(1, 2), head + "bias", F.conv1d).transpose(1, 2)
        want = torch.log_softmax(lg, -1)[0]
        got = setup["lp"][b, :t].cpu()
        _report(f"head[{b}]", got, want, 2e-6)


def test_logprobs_against_the_oracle_and_its_noise_floor(setup):
    R, w, T = setup["R"], setup["w"], setup["t"]
    lp_ref, t_ref = R.forward(w, setup["audio"], LENS, ort=R.OrtMixed())
    assert t_ref.tolist() == T
    got = setup["lp"].cpu()
    d = torch.cat([(got[b, : T[b]] - lp_ref[b, : T[b]]).flatten() for b in range(len(T))])
    mx, rms = float(d.abs().max()), float(d.pow(2).mean().sqrt())
    same = sum(int((got[b, : T[b]].argmax(-1) == lp_ref[b, : T[b]].argmax(-1)).sum()) for b in range(len(T))) / sum(T)
    print(f"[ort-e2e] hip_ort vs OrtMixed: max {mx:.4f} rms {rms:.5f} argmax {same:.4f}")
    # the oracle against itself (threads 1 vs 16): 0.05-0.07 max, 0.012 rms, 0.97-0.99 argmax
    assert mx <= 0.2 and rms <= 0.03 and same >= 0.95, (mx, rms, same)
    for b, n in enumerate(T):
        assert torch.allclose(got[b, :n].exp().sum(-1), torch.ones(n), atol=1e-4)
    # the f16-arithmetic mixed path is a different model from this one
    from offline_tarteel_amd.engine import Engine

    eng1 = Engine(device=0, with_model=True, seed=SEED, precision=1, max_batch=4, max_samples=80000)
    try:
        lp1, _ = eng1.forward(setup["audio"].cuda().contiguous(), LENS)
        d1 = torch.cat([(lp1[b, : T[b]].cpu() - lp_ref[b, : T[b]]).flatten() for b in range(len(T))])
        print(f"[ort-e2e] hip_mixed (W4A16/W8A16) vs OrtMixed: max {float(d1.abs().max()):.4f} rms {float(d1.pow(2).mean().sqrt()):.5f}")
        assert float(d1.pow(2).mean().sqrt()) > 1.5 * rms
    finally:
        eng1.close()


def test_batch_invariance_is_exact(setup):
    """the activation ranges are per utterance: an utterance alone == the same utterance inside a ragged batch, bit for bit
    (the reference quantises per call and feeds batch 1)."""
    eng, audio = setup["eng"], setup["audio"]
    for b in (1, 2):
        n = LENS[b]
        lp1, t1 = eng.forward(audio[b: b + 1, :n].cuda().contiguous(), [n])
        assert t1[0] == setup["t"][b]
        assert torch.equal(lp1[0, : t1[0]], setup["lp"][b, : t1[0]])


def test_predict_batch_runs_the_whole_path(setup, oracle):
    eng = setup["eng"]
    res = eng.predict_batch(setup["audio"].cuda().contiguous(), LENS)
    for i, n in enumerate(setup["t"]):
        want = oracle.predict_logprobs(setup["lp"][i, :n].cpu().numpy())
        assert res[i]["greedy_ids"] == want["greedy_ids"]
        assert (res[i]["surah"], res[i]["ayah"], res[i]["ayah_end"], res[i]["source"]) == (
            want["surah"], want["ayah"], want["ayah_end"], want["source"])


def test_configs2_batch256_ort_mixed_vs_oracle_sample():
    """BASELINE configs[2] at full size (256 clips x 10 s) in the reference's arithmetic: two utterances of the batch
    against the oracle (noise-floor bound), and a third against itself run alone (exact)."""
    from offline_tarteel_amd.engine import Engine
    from oracle import fastconformer_ref as R

    n = 160000
    audio = torch.from_numpy(synth_audio(256, n, seed=31))
    eng = Engine(device=0, with_model=True, seed=20260630, precision=2, max_batch=256, max_samples=n)
    try:
        a = audio.cuda().contiguous()
        lp, T = eng.forward(a, [n] * 256)
        torch.cuda.synchronize()
        assert T == [126] * 256
        one, _ = eng.forward(a[200:201].contiguous(), [n])
        assert torch.equal(one[0], lp[200])
        w = R.random_weights(20260630)
        for b in (0, 255):
            ref, _ = R.forward(w, audio[b: b + 1], [n], ort=R.OrtMixed())
            d = (lp[b].cpu() - ref[0])
            mx, rms = float(d.abs().max()), float(d.pow(2).mean().sqrt())
            same = float((lp[b].cpu().argmax(-1) == ref[0].argmax(-1)).float().mean())
            print(f"[ort-e2e] B=256 utt {b}: max {mx:.4f} rms {rms:.5f} argmax {same:.4f}")
            assert mx <= 0.25 and rms <= 0.035 and same >= 0.93, (b, mx, rms, same)
        res = eng.predict_batch(a, [n] * 256, want_text=False)
        assert len(res) == 256 and all(r["t_frames"] == 126 for r in res)
    finally:
        eng.close()


def test_prequantised_file_in_the_export_form_runs_on_its_own_integers(tmp_path):
    """VERDICT r2 items 1(c) + 8: a full-size model written the way the reference's export is described (torch-export
    scopes, MatMulNBits WITH zero points, DynamicQuantizeLinear -> ConvInteger chains, STFT baked in; tests/synth_onnx.py)
    -> tools/convert_weights.py --onnx -> precision 2.  The file is marked pre-quantised: Linear weights run as the
    dequantised MatMulNBits values, Conv weights go back on the file's integers with the file's scale.  Checked against
    the oracle given the same dequantised values and scales; the same values WITHOUT the marker are re-quantised onto the
    engine's own symmetric int4 grid and must land measurably further away."""
    import importlib.util
    import sys
    from pathlib import Path

    from offline_tarteel_amd.engine import Engine
    from oracle import fastconformer_ref as R
    from synth_onnx import export_like_model

    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("convert_weights", str(root / "tools" / "convert_weights.py"))
    C = importlib.util.module_from_spec(spec)
    sys.modules["convert_weights"] = C
    spec.loader.exec_module(C)
    lib = C._lib()
    shapes = C.weight_shapes(lib)
    w0 = C.random_weights(lib, shapes, SEED)
    onnx_path = tmp_path / "export_like.onnx"
    deq, scales = export_like_model(w0, onnx_path)
    sd, meta = C.onnx_state_dict(str(onnx_path), shapes, verbose=False, with_meta=True)
    marked, plain = tmp_path / "marked.qvw", tmp_path / "plain.qvw"
    C.write_qvw(marked, {k: sd[k] for k in shapes}, C.prequantised_extras(meta))
    C.write_qvw(plain, {k: sd[k] for k in shapes})
    wt = {k: torch.from_numpy(np.ascontiguousarray(deq[k], dtype=np.float32).reshape(shapes[k])) for k in shapes}
    lens = [32000, 17777]
    audio = torch.from_numpy(synth_audio(2, 32000))
    audio[1, lens[1]:] = 0
    torch.set_num_threads(min(16, torch.get_num_threads()))
    lp_ref, t_ref = R.forward(wt, audio, lens, ort=R.OrtMixed(int4_linears=False, conv_scales=scales))
    T = t_ref.tolist()

    def run(path):
        eng = Engine(device=0, with_model=True, weights_path=str(path), precision=2, max_batch=2, max_samples=32000)
        try:
            lp, t = eng.forward(audio.cuda().contiguous(), lens)
            assert t == T
            d = torch.cat([(lp[b, : T[b]].cpu() - lp_ref[b, : T[b]]).flatten() for b in range(len(T))])
            return float(d.abs().max()), float(d.pow(2).mean().sqrt())
        finally:
            eng.close()

    mx, rms = run(marked)
    mx_p, rms_p = run(plain)
    # precisions 0 and 1 on a marked file: nothing is re-quantised, so both run the file's values as f16 operands -- the
    # same bits (precision 1 on an UNMARKED file packs its own int4 / int8 grids and differs)
    outs = {}
    for prec, path in ((0, marked), (1, marked), (1, plain)):
        eng = Engine(device=0, with_model=True, weights_path=str(path), precision=prec, max_batch=2, max_samples=32000)
        try:
            lp, t = eng.forward(audio.cuda().contiguous(), lens)
            outs[(prec, path.name)] = lp.cpu()
        finally:
            eng.close()
    def valid(lp):   # the frames that exist; rows t >= T[b] are qv_forward's zero padding (checked below)
        return torch.cat([lp[b, : T[b]].flatten() for b in range(len(T))])

    assert torch.equal(valid(outs[(0, "marked.qvw")]), valid(outs[(1, "marked.qvw")]))
    assert not torch.equal(valid(outs[(1, "plain.qvw")]), valid(outs[(1, "marked.qvw")]))
    for lp in outs.values():
        assert all(bool((lp[b, T[b]:] == 0).all()) for b in range(len(T)))
    print(f"[ort-prequant] marked file vs oracle: max {mx:.4f} rms {rms:.5f}; unmarked (re-quantised): max {mx_p:.4f} rms {rms_p:.5f}")
    assert mx <= 0.2 and rms <= 0.03, (mx, rms)
    assert rms_p > 1.5 * rms, (rms_p, rms)


def test_edge_shapes_one_frame_thirty_seconds_and_silence():
    """precision 2 on the shapes the fp16 path is tested on: the shortest legal clip (400 samples -> 1 encoder frame:
    every per-utterance range is taken over a handful of values), a 30 s clip (T = 376: several query groups, the long
    CTC kernel), an all-zero clip (degenerate ranges: scale 0 -> the ONNX rule x_q = 0, zero point 0) and a near-silent
    one -- ragged in one batch, against each utterance run alone (exact) and against the oracle (noise-floor bound).
    Exact digital silence is NOT compared with the oracle: every mel frame is the same constant, the per-feature std is 0
    and (x - mean) / (0 + 1e-5) amplifies the rounding of the MEAN by 1e5 -- zero here (f64 statistics), O(1) noise in
    the oracle's float32 sum; no two implementations agree there, the reference's included."""
    from offline_tarteel_amd.engine import Engine
    from oracle import fastconformer_ref as R

    lens = [400, 480000, 16000, 48000, 16000]
    audio = torch.from_numpy(synth_audio(5, 480000, seed=91))
    for b, n in enumerate(lens):
        audio[b, n:] = 0
    audio[2] = 0                                         # silence
    audio[4, :16000] *= 1e-3                             # near silence
    eng = Engine(device=0, with_model=True, seed=SEED, precision=2, max_batch=5, max_samples=480000)
    try:
        a = audio.cuda().contiguous()
        lp, T = eng.forward(a, lens)
        torch.cuda.synchronize()
        assert T == [1, 376, 13, 38, 13]
        assert all(bool(torch.isfinite(lp[b, : T[b]]).all()) for b in range(5))
        for b in range(5):
            one, t1 = eng.forward(a[b: b + 1, : lens[b]].contiguous(), [lens[b]])
            assert t1[0] == T[b] and torch.equal(one[0, : T[b]], lp[b, : T[b]]), b
        w = R.random_weights(SEED)
        torch.set_num_threads(min(16, torch.get_num_threads()))
        for b in (0, 4, 3, 1):
            ref, tr = R.forward(w, audio[b: b + 1, : lens[b]].contiguous(), [lens[b]], ort=R.OrtMixed())
            assert int(tr[0]) == T[b]
            d = lp[b, : T[b]].cpu() - ref[0, : T[b]]
            mx, rms = float(d.abs().max()), float(d.pow(2).mean().sqrt())
            print(f"[ort-edge] utt {b} (T = {T[b]}): max {mx:.4f} rms {rms:.5f}")
            assert mx <= 0.25 and rms <= 0.04, (b, mx, rms)
        res = eng.predict_batch(a, lens)
        assert [r["t_frames"] for r in res] == T
    finally:
        eng.close()
