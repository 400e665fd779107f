"""Synthetic ONNX models for the converter tests: a small protobuf writer (independent of tools/onnx_reader.py -- it
only shares the public onnx.proto field numbers) and `export_like_model`, which writes a weight dict in the form the
reference's fastconformer_full_mixed.onnx is described to have (experiments/c2c-direct-mixed/run.py:1-9,
web/frontend/public/export_metadata.json): torch-export node scopes ("/encoder/layers.N/feed_forward1/linear1/..."),
anonymous "onnx::MatMul_N" operands turned into MatMulNBits int4 blocks WITH zero points, every Conv as
DynamicQuantizeLinear -> ConvInteger -> Cast -> Mul -> Mul -> Add(anonymous bias) on "<name>_quantized / _scale /
_zero_point" initializers, float LayerNorm / BatchNorm / pos_bias tensors, and the STFT / mel front-end baked into the
graph as further nodes and initializers the converter must ignore.  [The real file is absent; this is its description.]"""

import numpy as np


# ------------------------------------------------------------------ tiny protobuf writer --
def vint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def key(num, wt):
    return vint(num << 3 | wt)


def ld(num, payload):
    return key(num, 2) + vint(len(payload)) + payload


def tensor(name, arr, how="raw"):
    code = {np.dtype(np.float32): 1, np.dtype(np.uint8): 2, np.dtype(np.int8): 3, np.dtype(np.float16): 10,
            np.dtype(np.int64): 7}[arr.dtype]
    out = b"".join(key(1, 0) + vint(d) for d in arr.shape) + key(2, 0) + vint(code)
    if how == "raw":
        out += ld(9, arr.astype(arr.dtype.newbyteorder("<")).tobytes())
    elif how == "float_data":                       # packed repeated float
        out += ld(4, arr.astype("<f4").tobytes())
    elif how == "int32_data":                       # fp16 bit patterns / small ints, packed varints
        vals = arr.view(np.uint16).ravel() if arr.dtype == np.float16 else arr.ravel()
        out += ld(5, b"".join(vint(int(v)) for v in vals))
    return out + ld(8, name.encode())


def attr_t(name, tensor_bytes):
    return ld(1, name.encode()) + ld(5, tensor_bytes) + key(20, 0) + vint(4)


def attr_i(name, v):
    return ld(1, name.encode()) + key(3, 0) + vint(v) + key(20, 0) + vint(2)


def node(op, name, inputs, outputs, attrs=()):
    out = b"".join(ld(1, i.encode()) for i in inputs) + b"".join(ld(2, o.encode()) for o in outputs)
    out += ld(3, name.encode()) + ld(4, op.encode())
    return out + b"".join(ld(5, a) for a in attrs)


def model(nodes, inits):
    graph = b"".join(ld(1, n) for n in nodes) + ld(2, b"g") + b"".join(ld(5, t) for t in inits)
    return key(1, 0) + vint(8) + ld(2, b"test") + ld(7, graph)


def pack_nbits(w, bs, with_zp):
    """block-wise 4-bit quantisation in MatMulNBits layout; returns (B, scales, zp_packed | None, dequantised)."""
    N, K = w.shape
    nb = K // bs
    blocks = w.reshape(N, nb, bs)
    if with_zp:
        lo, hi = blocks.min(-1, keepdims=True), blocks.max(-1, keepdims=True)
        scale = np.maximum((hi - lo) / 15.0, 1e-8).astype(np.float32)
        zp = np.clip(np.rint(-lo / scale), 0, 15)
    else:
        scale = np.maximum(np.abs(blocks).max(-1, keepdims=True) / 7.0, 1e-8).astype(np.float32)
        zp = np.full_like(scale, 8.0)
    q = np.clip(np.rint(blocks / scale + zp), 0, 15).astype(np.uint8)
    B = (q[:, :, 0::2] | (q[:, :, 1::2] << 4)).astype(np.uint8)
    deq = ((q.astype(np.float32) - zp) * scale).reshape(N, K)
    zpp = None
    if with_zp:
        z = zp.reshape(N, nb).astype(np.uint8)
        if nb % 2:
            z = np.concatenate([z, np.zeros((N, 1), np.uint8)], 1)
        zpp = (z[:, 0::2] | (z[:, 1::2] << 4)).astype(np.uint8)
    return B, scale.reshape(-1), zpp, deq


# ------------------------------------------------------------------ a model shaped like the reference's export ----
LINEAR_TAILS = ("feed_forward1.linear1.weight", "feed_forward1.linear2.weight", "feed_forward2.linear1.weight",
                "feed_forward2.linear2.weight", "self_attn.linear_q.weight", "self_attn.linear_k.weight",
                "self_attn.linear_v.weight", "self_attn.linear_out.weight", "self_attn.linear_pos.weight",
                "encoder.pre_encode.out.weight")


def _scope_parts(module: str):
    """'encoder.layers.3.conv.pointwise_conv1' -> ['encoder', 'layers.3', 'conv', 'pointwise_conv1'] (torch export
    keeps a ModuleList / Sequential index glued to its container name)."""
    parts, out = module.split("."), []
    for p in parts:
        if p.isdigit() and out:
            out[-1] = out[-1] + "." + p
        else:
            out.append(p)
    return out


def export_like_model(weights: dict, path, block: int = 128):
    """weights: {NeMo key: float32 array}.  Writes `path`; returns (dequantised {key: array}, {conv weight key: float32
    scale}) -- what the file holds once its integers are multiplied out."""
    nodes, inits, deq, scales = [], [], {}, {}
    uid = [1000]

    def anon(kind):
        uid[0] += 1
        return f"onnx::{kind}_{uid[0]}"

    # the pre-processor baked into the graph (export_metadata.json: STFT + mel inside the model): nothing of it is a weight
    inits.append(tensor("preprocessor.featurizer.window", np.hanning(400).astype(np.float32)))
    inits.append(tensor("preprocessor.featurizer.fb", np.zeros((1, 80, 257), np.float32)))
    inits.append(tensor(anon("Reshape"), np.array([0, -1, 80], np.int64)))
    nodes.append(node("STFT", "/preprocessor/featurizer/STFT", ["audio_signal", "frame_step", "preprocessor.featurizer.window"],
                      ["stft"], [attr_i("onesided", 1)]))
    nodes.append(node("MatMul", "/preprocessor/featurizer/MatMul", ["preprocessor.featurizer.fb", "power"], ["mel"]))

    biases_behind_add = {}
    for name, w in weights.items():
        w = np.asarray(w, np.float32)
        if name == "ctc_decoder.decoder_layers.0.weight" and w.ndim == 2:
            w = w[:, :, None]                           # the CTC head is a Conv1d(512, 1025, 1)
        module = name.rsplit(".", 1)[0]
        scope = "/" + "/".join(_scope_parts(module))
        if name.endswith(LINEAR_TAILS):
            B, s, z, d = pack_nbits(w, block, with_zp=True)
            base = anon("MatMul")
            nodes.append(node("MatMulNBits", scope + "/MatMul_Q4", [scope + "/in", base + "_Q4", base + "_scales", base + "_zero_points"],
                              [scope + "/MatMul_output_0"],
                              [attr_i("K", w.shape[1]), attr_i("N", w.shape[0]), attr_i("bits", 4), attr_i("block_size", block)]))
            inits += [tensor(base + "_Q4", B), tensor(base + "_scales", s), tensor(base + "_zero_points", z)]
            deq[name] = d
        elif name.endswith(".weight") and w.ndim >= 3:
            sw = np.float32(float(np.abs(w).max()) / 127.0)
            q = np.clip(np.rint(w / sw), -127, 127).astype(np.int8)
            bias_name = anon("Add")
            nodes += [
                node("DynamicQuantizeLinear", scope + "/DynamicQuantizeLinear", [scope + "/in"], [scope + "/in_q", scope + "/in_s", scope + "/in_z"]),
                node("ConvInteger", scope + "/Conv_quant", [scope + "/in_q", name + "_quantized", scope + "/in_z", name + "_zero_point"],
                     [scope + "/acc"]),
                node("Cast", scope + "/Cast", [scope + "/acc"], [scope + "/acc_f"], [attr_i("to", 1)]),
                node("Mul", scope + "/Mul_scales", [scope + "/in_s", name + "_scale"], [scope + "/s"]),
                node("Mul", scope + "/Mul", [scope + "/acc_f", scope + "/s"], [scope + "/scaled"]),
                node("Add", scope + "/Add", [scope + "/scaled", bias_name], [scope + "/Conv_output_0"]),
            ]
            inits += [tensor(name + "_quantized", q), tensor(name + "_scale", np.array(sw, np.float32)),
                      tensor(name + "_zero_point", np.array(0, np.int8))]
            biases_behind_add[module + ".bias"] = (bias_name, w.shape[0], w.ndim)
            deq[name] = q.astype(np.float32) * sw
            scales[name] = sw
        elif name in biases_behind_add or (name.endswith(".bias") and module + ".weight" in weights and
                                           (np.asarray(weights[module + ".weight"]).ndim >= 3 or module.startswith("ctc_decoder"))):
            continue                                    # written below, anonymously, behind its ConvInteger chain
        else:
            inits.append(tensor(name, w))
            deq[name] = w
    for bname, (anon_name, C, nd) in biases_behind_add.items():
        b = np.asarray(weights[bname], np.float32)
        inits.append(tensor(anon_name, b.reshape((1, C) + (1,) * (nd - 2))))
        deq[bname] = b
    with open(path, "wb") as f:
        f.write(model(nodes, inits))
    return deq, scales
