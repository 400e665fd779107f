"""tools/onnx_reader.py + the --onnx leg of tools/convert_weights.py on synthetic ONNX models.

The models are written by the small protobuf encoder in tests/synth_onnx.py (independent of the reader: it only
shares the public onnx.proto field numbers), in the storage forms the reference's
fastconformer_full_mixed.onnx is described to use: MatMulNBits int4 blocks (with default and with
packed zero points), int8 tensors behind DequantizeLinear, ConvInteger weights written by dynamic
quantisation, plain float initializers in raw / typed / fp16 encodings, and an anonymous MatMul
operand that only the consuming node's scope identifies."""

import pytest
import importlib.util
import struct
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def _load(name):
    spec = importlib.util.spec_from_file_location(name, str(ROOT / "tools" / f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


from synth_onnx import attr_i, attr_t, key, ld, model, node, pack_nbits, tensor, vint  # noqa: E402,F401  (the protobuf writer)


def test_wire_primitives_and_tensor_encodings(tmp_path):
    O = _load("onnx_reader")
    a = np.arange(12, dtype=np.float32).reshape(3, 4) - 5
    h = (np.arange(6, dtype=np.float32) / 4 - 1).astype(np.float16).reshape(2, 3)
    i64 = np.array([-1, 2 ** 40, 7], np.int64)
    p = tmp_path / "m.onnx"
    p.write_bytes(model([], [tensor("a_raw", a), tensor("a_typed", a, "float_data"), tensor("h", h, "int32_data"),
                             key(1, 0) + vint(3) + key(2, 0) + vint(7) +          # int64_data, packed varints
                             ld(7, b"".join(vint(int(v)) for v in i64)) + ld(8, b"i64")]))
    nodes, inits = O.read_model(p)
    assert nodes == []
    assert np.array_equal(inits["a_raw"], a) and np.array_equal(inits["a_typed"], a)
    assert inits["h"].dtype == np.float16 and np.array_equal(inits["h"], h)
    assert inits["i64"].tolist() == i64.tolist()


def test_matmul_nbits_known_answer():
    O = _load("onnx_reader")
    q = np.arange(16, dtype=np.uint8)                        # element 2i in the low nibble
    B = (q[0::2] | (q[1::2] << 4)).reshape(1, 1, 8)
    n = O.Node(op="MatMulNBits", name="/x/MatMul_Q4", inputs=["a", "B", "s"], outputs=["y"],
               attrs={"K": 16, "N": 1, "bits": 4, "block_size": 16})
    w = O.dequant_matmul_nbits(n, {"B": B, "s": np.array([0.5], np.float32)})
    assert np.array_equal(w, ((q.astype(np.float32) - 8) * 0.5).reshape(1, 16))   # default zero point 8


def test_convert_weights_from_synthetic_onnx(tmp_path):
    C = _load("convert_weights")
    rng = np.random.default_rng(3)
    D, FF, V = 32, 64, 11
    L = "encoder.layers.0."
    shapes = {
        L + "feed_forward1.linear1.weight": (FF, D), L + "feed_forward1.linear1.bias": (FF,),
        L + "feed_forward1.linear2.weight": (D, FF),
        L + "self_attn.linear_pos.weight": (D, D),
        L + "conv.pointwise_conv1.weight": (2 * D, D, 1),
        L + "conv.depthwise_conv.weight": (D, 1, 9),
        L + "conv.batch_norm.weight": (D,), L + "conv.batch_norm.bias": (D,),
        L + "conv.batch_norm.running_mean": (D,), L + "conv.batch_norm.running_var": (D,),
        L + "norm_out.weight": (D,), L + "norm_out.bias": (D,),
        "ctc_decoder.decoder_layers.0.weight": (V, D, 1),
    }
    w1 = rng.normal(size=(FF, D)).astype(np.float32)
    w2 = rng.normal(size=(D, FF)).astype(np.float32)
    B1, s1, _, deq1 = pack_nbits(w1, 16, with_zp=False)
    B2, s2, z2, deq2 = pack_nbits(w2, 32, with_zp=True)
    b1 = rng.normal(size=FF).astype(np.float32)
    wpos = rng.normal(size=(D, D)).astype(np.float32)                 # [out, in]; stored transposed for MatMul
    pw1 = rng.normal(size=(2 * D, D, 1)).astype(np.float32)
    pw_scale = np.float32(np.abs(pw1).max() / 127.0)
    pw_q = np.clip(np.rint(pw1 / pw_scale) + 128, 0, 255).astype(np.uint8)
    pw_deq = (pw_q.astype(np.float32) - 128.0) * pw_scale
    dw = rng.normal(size=(D, 1, 9)).astype(np.float32)
    g = (1 + 0.1 * rng.normal(size=D)).astype(np.float32)
    g_scale = np.float32(np.abs(g).max() / 127.0)
    g_q = np.clip(np.rint(g / g_scale), -127, 127).astype(np.int8)
    beta = rng.normal(size=D).astype(np.float16)
    head = rng.normal(size=(V, D, 1)).astype(np.float32)
    nodes = [
        node("MatMulNBits", "/encoder/layers.0/feed_forward1/linear1/MatMul_Q4", ["x", "onnx::MatMul_901_Q4", "onnx::MatMul_901_scales"],
             ["y1"], [attr_i("K", D), attr_i("N", FF), attr_i("bits", 4), attr_i("block_size", 16)]),
        node("MatMulNBits", "/encoder/layers.0/feed_forward1/linear2/MatMul_Q4",
             ["y1", "onnx::MatMul_902_Q4", "onnx::MatMul_902_scales", "onnx::MatMul_902_zero_points"], ["y2"],
             [attr_i("K", FF), attr_i("N", D), attr_i("bits", 4), attr_i("block_size", 32)]),
        node("MatMul", "/layers.0/self_attn/linear_pos/MatMul", ["pos", "onnx::MatMul_903"], ["p"]),   # scope without "encoder."
        node("ConvInteger", "/encoder/layers.0/conv/pointwise_conv1/Conv_quant",
             ["xq", L + "conv.pointwise_conv1.weight_quantized", "x_zp", L + "conv.pointwise_conv1.weight_zero_point"], ["c"]),
        node("DequantizeLinear", "/encoder/layers.0/norm_out/dq", [L + "norm_out.weight_quantized", "g_scale", "g_zp"],
             [L + "norm_out.weight_dq"]),
    ]
    inits = [
        tensor("onnx::MatMul_901_Q4", B1), tensor("onnx::MatMul_901_scales", s1),
        tensor("onnx::MatMul_902_Q4", B2), tensor("onnx::MatMul_902_scales", s2), tensor("onnx::MatMul_902_zero_points", z2),
        tensor(L + "feed_forward1.linear1.bias", b1, "float_data"),
        tensor("onnx::MatMul_903", np.ascontiguousarray(wpos.T)),
        tensor(L + "conv.pointwise_conv1.weight_quantized", pw_q), tensor(L + "conv.pointwise_conv1.weight_scale", np.array(pw_scale)),
        tensor(L + "conv.pointwise_conv1.weight_zero_point", np.array(128, np.uint8)),
        tensor(L + "conv.depthwise_conv.weight", dw),
        tensor(L + "norm_out.weight_quantized", g_q), tensor("g_scale", np.array(g_scale)), tensor("g_zp", np.array(0, np.int8)),
        tensor(L + "norm_out.bias", beta, "int32_data"),
        tensor("ctc_decoder.decoder_layers.0.weight", head),
    ]
    p = tmp_path / "synthetic.onnx"
    p.write_bytes(model(nodes, inits))
    sd = C.onnx_state_dict(str(p), shapes, verbose=False)
    assert set(sd) == set(shapes)
    assert np.array_equal(sd[L + "feed_forward1.linear1.weight"], deq1)
    assert np.array_equal(sd[L + "feed_forward1.linear2.weight"], deq2)
    assert np.abs(deq2 - w2).max() < 0.25 and np.abs(deq1 - w1).max() < 0.5      # the synthetic quantiser is sane
    assert np.array_equal(sd[L + "feed_forward1.linear1.bias"], b1)
    assert np.array_equal(sd[L + "self_attn.linear_pos.weight"], wpos)           # MatMul operand transposed back
    assert np.array_equal(sd[L + "conv.pointwise_conv1.weight"], pw_deq)
    assert np.array_equal(sd[L + "conv.depthwise_conv.weight"], dw)
    assert np.array_equal(sd[L + "norm_out.weight"], g_q.astype(np.float32) * g_scale)
    assert np.array_equal(sd[L + "norm_out.bias"], beta.astype(np.float32))
    assert np.array_equal(sd["ctc_decoder.decoder_layers.0.weight"], head)
    # BatchNorm absent from the graph -> identity: (x - 0) / sqrt(var + 1e-5) * 1 + 0 == x
    assert np.all(sd[L + "conv.batch_norm.weight"] == 1) and np.all(sd[L + "conv.batch_norm.running_mean"] == 0)
    assert np.allclose(sd[L + "conv.batch_norm.running_var"] + 1e-5, 1.0)
    # and the flat file the engine loads
    out = tmp_path / "w.qvw"
    C.write_qvw(out, {k: sd[k] for k in shapes})
    raw = out.read_bytes()
    assert raw[:8] == b"QVWT0001" and struct.unpack("<I", raw[8:12])[0] == len(shapes)
    # a tensor the graph does not hold is reported, not guessed
    import pytest

    with pytest.raises(SystemExit):
        C.onnx_state_dict(str(p), dict(shapes, **{L + "self_attn.linear_q.weight": (D, D)}), verbose=False)


def test_constant_nodes_gemm_and_anonymous_conv_operands(tmp_path):
    """Weights that are not plainly named initializers: a Constant node, a Gemm with transB = 1 under
    a prefixed scope, and a Conv whose weight and bias lost their names to BatchNorm folding."""
    C = _load("convert_weights")
    rng = np.random.default_rng(4)
    D = 16
    L = "encoder.layers.0."
    shapes = {L + "self_attn.linear_q.weight": (D, D), L + "self_attn.linear_q.bias": (D,),
              L + "conv.depthwise_conv.weight": (D, 1, 9), L + "conv.depthwise_conv.bias": (D,),
              L + "norm_conv.weight": (D,)}
    wq, bq = rng.normal(size=(D, D)).astype(np.float32), rng.normal(size=D).astype(np.float32)
    dw, db = rng.normal(size=(D, 1, 9)).astype(np.float32), rng.normal(size=D).astype(np.float32)
    g = rng.normal(size=D).astype(np.float32)
    nodes = [
        node("Constant", "/c0", [], [L + "norm_conv.weight"], [attr_t("value", tensor("", g))]),
        node("Gemm", "/model/encoder/layers.0/self_attn/linear_q/Gemm", ["x", "onnx::Gemm_11", "onnx::Gemm_12"], ["q"],
             [attr_i("transB", 1)]),
        node("Conv", "/model/encoder/layers.0/conv/depthwise_conv/Conv", ["y", "onnx::Conv_21", "onnx::Conv_22"], ["z"]),
    ]
    inits = [tensor("onnx::Gemm_11", wq), tensor("onnx::Gemm_12", bq), tensor("onnx::Conv_21", dw), tensor("onnx::Conv_22", db)]
    p = tmp_path / "m.onnx"
    p.write_bytes(model(nodes, inits))
    sd = C.onnx_state_dict(str(p), shapes, verbose=False)
    assert np.array_equal(sd[L + "self_attn.linear_q.weight"], wq) and np.array_equal(sd[L + "self_attn.linear_q.bias"], bq)
    assert np.array_equal(sd[L + "conv.depthwise_conv.weight"], dw) and np.array_equal(sd[L + "conv.depthwise_conv.bias"], db)
    assert np.array_equal(sd[L + "norm_conv.weight"], g)


def test_dynamic_quantised_conv_with_bias_behind_add_and_valueless_attribute(tmp_path):
    """What onnxruntime's quantize_dynamic leaves of a biased Conv under a real export's node naming
    (/encoder/layers.N/...): ConvInteger on an int8 weight -> Cast -> Mul(scale) -> Add(anonymous bias).
    The weight is found under '<scope>.weight', the bias by following the consumer chain to the Add.
    Also: an attribute serialised without a value (proto3 drops transB = 0) parses as the scalar 0."""
    C = _load("convert_weights")
    R = _load("onnx_reader")
    rng = np.random.default_rng(5)
    D = 8
    L = "encoder.layers.3."
    shapes = {L + "conv.pointwise_conv2.weight": (D, D, 1), L + "conv.pointwise_conv2.bias": (D,),
              L + "feed_forward1.linear1.weight": (D, D), L + "feed_forward1.linear1.bias": (D,)}
    wq = rng.integers(-127, 128, size=(D, D, 1)).astype(np.int8)
    scale = np.float32(0.013)
    bias = rng.normal(size=D).astype(np.float32)
    gw, gb = rng.normal(size=(D, D)).astype(np.float32), rng.normal(size=D).astype(np.float32)
    scope = "/encoder/layers.3/conv/pointwise_conv2/"
    valueless = ld(1, b"transB") + key(20, 0) + vint(2)            # AttributeProto{name, type=INT}, i omitted
    nodes = [
        node("DynamicQuantizeLinear", scope + "DQL", ["h"], ["h_q", "h_s", "h_z"]),
        node("ConvInteger", scope + "Conv_quant", ["h_q", "pw2_quantized", "h_z", "pw2_zero_point"], ["acc"]),
        node("Cast", scope + "Cast", ["acc"], ["acc_f"]),
        node("Mul", scope + "Mul", ["acc_f", "mul_scales"], ["scaled"]),
        node("Add", scope + "Add", ["scaled", "onnx::Add_77"], ["out"]),
        node("Gemm", "/encoder/layers.3/feed_forward1/linear1/Gemm", ["x", "onnx::Gemm_5", "onnx::Gemm_6"], ["y"], [valueless]),
    ]
    inits = [tensor("pw2_quantized", wq), tensor("pw2_scale", np.array([scale], np.float32)),
             tensor("pw2_zero_point", np.zeros(1, np.int8)), tensor("mul_scales", np.array([0.5], np.float32)),
             tensor("onnx::Add_77", bias.reshape(1, D, 1)), tensor("onnx::Gemm_5", gw.T.copy()), tensor("onnx::Gemm_6", gb)]
    p = tmp_path / "dq.onnx"
    p.write_bytes(model(nodes, inits))
    ns, _ = R.read_model(str(p))
    assert [n for n in ns if n.op == "Gemm"][0].attrs["transB"] == 0
    sd = C.onnx_state_dict(str(p), shapes, verbose=False)
    assert np.array_equal(sd[L + "conv.pointwise_conv2.weight"], wq.astype(np.float32) * scale)
    assert np.array_equal(sd[L + "conv.pointwise_conv2.bias"], bias)
    # Gemm without transB holds [in, out]: transposed to the state-dict's [out, in]
    assert np.array_equal(sd[L + "feed_forward1.linear1.weight"], gw) and np.array_equal(sd[L + "feed_forward1.linear1.bias"], gb)


def test_export_like_model_keeps_the_files_own_quantisation(tmp_path):
    """A model written the way the reference's export is described (tests/synth_onnx.py::export_like_model: torch-export
    scopes, anonymous MatMulNBits operands WITH zero points, DynamicQuantizeLinear -> ConvInteger -> Cast -> Mul -> Add
    chains with the bias behind the Add, the STFT front-end baked in) through the converter: every tensor placed, the
    values are the file's integers multiplied out, the int8 scales travel verbatim, the file is marked pre-quantised,
    and the oracle's precision-2 routing puts a Conv weight back on exactly the file's integers."""
    import torch
    from synth_onnx import LINEAR_TAILS, export_like_model

    C = _load("convert_weights")
    lib = C._lib()
    full = C.weight_shapes(lib)
    w_all = C.random_weights(lib, full, 11)
    keep = [k for k in full if "encoder.layers." not in k or ".layers.0." in k or ".layers.16." in k]
    shapes = {k: full[k] for k in keep}
    p = tmp_path / "export_like.onnx"
    deq, scales = export_like_model({k: w_all[k] for k in keep}, p)
    sd, meta = C.onnx_state_dict(str(p), shapes, verbose=False, with_meta=True)
    assert set(sd) == set(shapes)
    for k in shapes:
        assert np.array_equal(sd[k], deq[k].reshape(shapes[k])), k
    lin = [k for k in shapes if k.endswith(LINEAR_TAILS)]
    conv = [k for k in shapes if k.endswith(".weight") and (len(shapes[k]) >= 3 or k.startswith("ctc_decoder"))]
    assert len(lin) == 2 * 9 + 1 and len(conv) == 5 + 2 * 3 + 1
    assert all(meta[k]["kind"] == "int4" for k in lin)
    assert all(meta[k]["kind"] == "int8" and meta[k]["scale"] == scales[k] and meta[k]["zero_point"] == 0 for k in conv)
    assert set(meta) == set(lin) | set(conv)
    # the int4 blocks really are asymmetric: the engine's own symmetric packing could not hold them
    from oracle import fastconformer_ref as R
    k0 = "encoder.layers.0.feed_forward1.linear1.weight"
    assert np.abs(R.quant_dequant_int4_f32scale(deq[k0]) - deq[k0]).max() > 1e-4
    extra = C.prequantised_extras(meta)
    # marker + one scale per int8 Conv weight + the file's own int4 grid (block scales and zero points) per Linear weight
    assert extra[C.PREQUANT_KEY].tolist() == [1.0] and len(extra) == 1 + len(conv) + 2 * len(lin)
    for k in lin:
        n_rows, kk = shapes[k][0], int(np.prod(shapes[k][1:]))
        gs, gz = extra[k + "#int4_scale"].reshape(n_rows, kk // 128), extra[k + "#int4_zp"].reshape(n_rows, kk // 128)
        q = sd[k].reshape(n_rows, kk // 128, 128) / gs[..., None] + gz[..., None]       # the file's integers, verbatim
        assert np.abs(q - np.rint(q)).max() <= 1e-3 and q.min() >= -1e-3 and q.max() <= 15 + 1e-3, k
    assert len({float(z) for k in lin for z in extra[k + "#int4_zp"]}) > 1                # asymmetric: zero points other than 8
    out = tmp_path / "w.qvw"
    C.write_qvw(out, {k: sd[k] for k in shapes}, extra)
    raw = out.read_bytes()
    assert struct.unpack("<I", raw[8:12])[0] == len(shapes) + len(extra)
    for k in conv:
        assert (k + C.SCALE_SUFFIX).encode() in raw
    # oracle: with the file's scale the weight integers come back verbatim (and without it they need not)
    ops = R.OrtMixed(int4_linears=False, conv_scales=scales)
    kc = "encoder.layers.0.conv.pointwise_conv1.weight"
    wt = {kc: torch.from_numpy(deq[kc])}
    ops.conv(wt, kc, torch.zeros(1, 512, 4), None, torch.nn.functional.conv1d)
    q_file = np.rint(deq[kc] / scales[kc])
    assert np.array_equal(ops._w8[kc][0].numpy(), q_file) and ops._w8[kc][1] == scales[kc]
    x = torch.ones(2, 512)
    assert torch.equal(ops.linear({k0: torch.from_numpy(deq[k0])}, k0, x, None), torch.nn.functional.linear(x, torch.from_numpy(deq[k0])))


def test_converter_refuses_int8_grids_the_engine_cannot_hold():
    """the engine's ConvInteger path holds ONE symmetric scale per Conv weight tensor.  A file quantised with
    per_channel=True or with a non-zero weight zero point must not be re-derived or clamped silently under a
    'runs on the file's own integers' marker: the converter refuses it.  MatMulInteger Linear weights get no scale entry
    (the engine reads none)."""
    C = _load("convert_weights")
    ok = {"a.conv.weight": {"kind": "int8", "op": "ConvInteger", "per_channel": False, "scale": np.float32(0.01), "zero_point": 0},
          "b.linear.weight": {"kind": "int8", "op": "MatMulInteger", "per_channel": False, "scale": np.float32(0.02), "zero_point": 0}}
    extra = C.prequantised_extras(ok)
    assert set(extra) == {C.PREQUANT_KEY, "a.conv.weight" + C.SCALE_SUFFIX}
    for bad in ({"per_channel": True, "zero_point": 0}, {"per_channel": False, "zero_point": 3}):
        meta = {"a.conv.weight": {"kind": "int8", "op": "ConvInteger", "scale": np.float32(0.01), **bad}}
        with pytest.raises(SystemExit):
            C.prequantised_extras(meta)
