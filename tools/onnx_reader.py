"""Minimal ONNX reader for the weight converter (no onnx / protobuf / onnxruntime needed).

Parses the protobuf wire format of an ONNX ModelProto far enough to get the graph's nodes
(op type, name, inputs, outputs, int/float/ints attributes) and its initializers as numpy arrays,
and dequantises the weight storage forms the reference's ``fastconformer_full_mixed.onnx`` is
described to use (README.md:22,41 of the reference: "int4 MatMul + int8 Conv/LayerNorm"):

* ``MatMulNBits`` (com.microsoft): B uint8 [N, K/block, block/2] with element 2i in the low
  nibble, scales float [N * K/block], optional packed uint8 zero points (default 8);
  W[n, k] = (q - zp) * scale, i.e. the [out, in] matrix of a Linear layer;
* ``DequantizeLinear`` on an initializer (per-tensor or per-axis scale / zero point);
* ``ConvInteger`` / ``MatMulInteger`` weights written by onnxruntime's dynamic quantisation
  (``<name>_quantized`` with ``<name>_scale`` and ``<name>_zero_point`` initializers).

Field numbers are those of onnx.proto (ModelProto.graph = 7; GraphProto.node = 1,
initializer = 5; NodeProto input = 1, output = 2, name = 3, op_type = 4, attribute = 5;
AttributeProto name = 1, f = 2, i = 3, s = 4, t = 5, floats = 7, ints = 8; TensorProto dims = 1,
data_type = 2, float_data = 4, int32_data = 5, int64_data = 7, name = 8, raw_data = 9).
The real file is not available in the build container, so this reader is exercised against
synthetic models written by tests/test_onnx_reader.py with an independent encoder.
"""

from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 4: np.uint16, 5: np.int16, 6: np.int32, 7: np.int64,
           10: np.float16, 11: np.float64, 12: np.uint32, 13: np.uint64}


def _varint(buf: bytes, pos: int):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def fields(buf: bytes):
    """Yield (field number, wire type, value) of one message; length-delimited values are bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield num, wt, val


def _signed(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


def _packed_varints(wt, val):
    if wt == 0:
        return [_signed(val)]
    out, pos = [], 0
    while pos < len(val):
        v, pos = _varint(val, pos)
        out.append(_signed(v))
    return out


def parse_tensor(buf: bytes):
    dims, dtype, name, raw = [], 1, "", None
    f32, i32, i64 = [], [], []
    for num, wt, val in fields(buf):
        if num == 1:
            dims += _packed_varints(wt, val)
        elif num == 2:
            dtype = val
        elif num == 4:
            f32.append(np.frombuffer(val, "<f4") if wt == 2 else np.frombuffer(val, "<f4", count=1))
        elif num == 5:
            i32 += _packed_varints(wt, val)
        elif num == 7:
            i64 += _packed_varints(wt, val)
        elif num == 8:
            name = val.decode()
        elif num == 9:
            raw = val
        elif num == 14 and val == 1:
            raise ValueError(f"tensor {name}: external data is not supported (re-save the model with embedded weights)")
    if dtype not in _DTYPES:
        raise ValueError(f"tensor {name}: unsupported data type {dtype}")
    np_dt = np.dtype(_DTYPES[dtype])
    if raw is not None:
        arr = np.frombuffer(raw, np_dt.newbyteorder("<"))
    elif f32:
        arr = np.concatenate(f32)
    elif i64:
        arr = np.array(i64, np.int64)
    elif dtype == 10:                       # float16 bit patterns travel in int32_data
        arr = np.array(i32, np.uint16).view(np.float16)
    else:
        arr = np.array(i32, np.int64)
    return name, arr.astype(np_dt, copy=False).reshape(dims if dims else (-1,) if arr.size != 1 else ())


@dataclass
class Node:
    op: str = ""
    name: str = ""
    inputs: list = field(default_factory=list)
    outputs: list = field(default_factory=list)
    attrs: dict = field(default_factory=dict)


def _parse_attr(buf: bytes):
    name, val = "", None
    floats, ints = [], []
    for num, wt, v in fields(buf):
        if num == 1:
            name = v.decode()
        elif num == 2:
            val = struct.unpack("<f", v)[0]
        elif num == 3:
            val = _signed(v)
        elif num == 4:
            val = v
        elif num == 5:
            val = parse_tensor(v)[1]
        elif num == 7:
            floats += list(np.frombuffer(v, "<f4")) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif num == 8:
            ints += _packed_varints(wt, v)
    if val is None:
        # repeated fields, or -- when nothing at all was serialised (proto3 omits default values, so
        # `transB = 0` arrives as a bare name) -- the scalar default 0, which int() / float() accept
        val = ints if ints else (floats if floats else 0)
    return name, val


def _parse_node(buf: bytes) -> Node:
    n = Node()
    for num, _wt, v in fields(buf):
        if num == 1:
            n.inputs.append(v.decode())
        elif num == 2:
            n.outputs.append(v.decode())
        elif num == 3:
            n.name = v.decode()
        elif num == 4:
            n.op = v.decode()
        elif num == 5:
            k, a = _parse_attr(v)
            n.attrs[k] = a
    return n


def read_model(path):
    """-> (nodes, initializers) of the main graph."""
    data = open(path, "rb").read()
    graph = None
    for num, wt, v in fields(data):
        if num == 7 and wt == 2:
            graph = v
    if graph is None:
        raise ValueError(f"{path}: no GraphProto (is this an ONNX model?)")
    nodes, inits = [], {}
    for num, wt, v in fields(graph):
        if num == 1 and wt == 2:
            nodes.append(_parse_node(v))
        elif num == 5 and wt == 2:
            name, arr = parse_tensor(v)
            inits[name] = arr
    return nodes, inits


# ------------------------------------------------------------------ dequantisation ----
def dequant_matmul_nbits(node: Node, inits: dict, with_grid: bool = False):
    """the [N, K] float32 matrix of a MatMulNBits node; with_grid: also (block size, scales [N, nb], zero points [N, nb])"""
    K, N = int(node.attrs["K"]), int(node.attrs["N"])
    bits, bs = int(node.attrs.get("bits", 4)), int(node.attrs["block_size"])
    if bits != 4:
        raise ValueError(f"{node.name}: MatMulNBits with bits = {bits} is not supported")
    nb = (K + bs - 1) // bs
    b = np.asarray(inits[node.inputs[1]], np.uint8).reshape(N, nb, bs // 2)
    q = np.empty((N, nb, bs), np.float32)
    q[:, :, 0::2] = b & 0x0F
    q[:, :, 1::2] = b >> 4
    scales = np.asarray(inits[node.inputs[2]], np.float32).reshape(N, nb, 1)
    if len(node.inputs) > 3 and node.inputs[3]:
        z = np.asarray(inits[node.inputs[3]])
        if z.dtype == np.uint8:
            z = z.reshape(N, -1)
            zp = np.empty((N, z.shape[1] * 2), np.float32)
            zp[:, 0::2] = z & 0x0F
            zp[:, 1::2] = z >> 4
            zp = zp[:, :nb].reshape(N, nb, 1)
        else:                                   # already unpacked (float zero points)
            zp = z.astype(np.float32).reshape(N, nb, 1)
    else:
        zp = np.full((N, nb, 1), 8.0, np.float32)
    w = ((q - zp) * scales).reshape(N, nb * bs)[:, :K]
    if with_grid:
        return w, bs, scales.reshape(N, nb).copy(), np.asarray(zp, np.float32).reshape(N, nb).copy()
    return w


def dequant_linear(x: np.ndarray, scale: np.ndarray, zp, axis: int = 1) -> np.ndarray:
    scale = np.asarray(scale, np.float32)
    zp = np.zeros_like(scale) if zp is None else np.asarray(zp, np.float32)
    if scale.ndim == 1 and scale.size > 1:
        shape = [1] * x.ndim
        shape[axis if axis >= 0 else x.ndim + axis] = -1
        scale, zp = scale.reshape(shape), zp.reshape(shape)
    return (x.astype(np.float32) - zp) * scale


def float_weights(nodes, inits):
    """Every weight-like tensor of the graph as float32, keyed by a name that identifies its
    module: {key: (array, how, meta)}; meta is None for float tensors, {"kind": "int4"} for MatMulNBits blocks and
    {"kind": "int8", "scale": float32, "zero_point": int} for a per-tensor int8 tensor (what onnxruntime's
    quantize_dynamic writes for Conv weights) -- the converter hands these to the engine so that its precision-2 path
    reproduces the file's integers instead of re-deriving a scale.  Keys are the initializer names for float initializers (with the
    quantisation suffixes stripped for dequantised ones) and additionally the scope of the
    consuming node ("/encoder/layers.0/feed_forward1/linear1/MatMul" -> "encoder.layers.0.
    feed_forward1.linear1") for anonymous MatMul operands."""
    out = {}
    inits = dict(inits)
    for n in nodes:                      # weights emitted as Constant nodes count as initializers
        if n.op == "Constant" and isinstance(n.attrs.get("value"), np.ndarray) and n.outputs:
            inits.setdefault(n.outputs[0], n.attrs["value"])
    for name, a in inits.items():
        if a.dtype in (np.float32, np.float16, np.float64):
            out[name] = (a.astype(np.float32), "float initializer", None)
    named = _named(out)
    for n in nodes:
        scope = n.name.strip("/").rsplit("/", 1)[0].replace("/", ".") if "/" in n.name.strip("/") else ""
        if n.op in ("Conv", "Gemm") and scope and len(n.inputs) > 1:
            # anonymous operands (e.g. "onnx::Conv_1234" after BatchNorm folding): name them by the module scope
            w = out.get(n.inputs[1])
            if w is not None and n.inputs[1] not in named:
                a = w[0]
                if n.op == "Gemm" and not int(n.attrs.get("transB", 0)):
                    a = a.T                                   # Gemm without transB holds [in, out]
                out[scope + ".weight"] = (a, f"{n.op} operand named by its node scope", w[2])
            if len(n.inputs) > 2 and n.inputs[2] in out and n.inputs[2] not in named:
                out[scope + ".bias"] = (out[n.inputs[2]][0], f"{n.op} bias named by its node scope", None)
        if n.op == "MatMulNBits":
            w, bs, g_scale, g_zp = dequant_matmul_nbits(n, inits, with_grid=True)
            for key in {scope, _strip(n.inputs[1])} - {""}:
                out[key] = (w, "MatMulNBits [out, in]", {"kind": "int4", "block": bs, "scales": g_scale, "zero_points": g_zp})
        elif n.op == "DequantizeLinear" and n.inputs[0] in inits:
            zp = inits.get(n.inputs[2]) if len(n.inputs) > 2 and n.inputs[2] else None
            w = dequant_linear(inits[n.inputs[0]], inits[n.inputs[1]], zp, int(n.attrs.get("axis", 1)))
            for key in {_strip(n.inputs[0]), n.outputs[0]}:
                out[key] = (w, "DequantizeLinear", None)
        elif n.op in ("ConvInteger", "MatMulInteger") and n.inputs[1] in inits:
            base = _strip(n.inputs[1])
            scale = inits.get(base + "_scale")
            if scale is None:
                continue
            zp = inits.get(n.inputs[3]) if len(n.inputs) > 3 and n.inputs[3] else inits.get(base + "_zero_point")
            w = dequant_linear(inits[n.inputs[1]], scale, zp, 0 if n.op == "ConvInteger" else 1)
            # (per-channel scales / non-zero weight zero points are carried so that the converter can REFUSE them: the
            # engine's precision-2 path holds one symmetric scale per Conv weight tensor, what quantize_dynamic writes)
            zps = np.asarray(zp).reshape(-1) if zp is not None else np.zeros(1)
            meta = {"kind": "int8", "op": n.op, "per_channel": bool(np.asarray(scale).size != 1),
                    "scale": np.float32(np.asarray(scale, np.float32).reshape(-1)[0]),
                    "zero_point": int(zps[0]) if zps.size == 1 else int(np.abs(zps).max())}
            keys = {base} | ({scope, scope + ".weight"} if scope else set())
            for key in keys:
                out[key] = (w if n.op == "ConvInteger" else w.T,
                            n.op + (" (transposed to [out, in])" if n.op != "ConvInteger" else ""), meta)
            # the bias of a dynamically quantised Conv / MatMul is an anonymous initializer added AFTER the
            # integer op and its rescaling: ConvInteger -> Cast -> Mul(scales) -> Add(bias).  Follow the single-
            # consumer chain of cheap elementwise nodes to that Add and name its constant operand.
            if scope:
                cur, hops = n.outputs[0], 0
                while hops < 6:
                    users = [m for m in nodes if cur in m.inputs]
                    if len(users) != 1:
                        break
                    u = users[0]
                    if u.op == "Add":
                        other = [i for i in u.inputs if i != cur]
                        if other and other[0] in inits and np.asarray(inits[other[0]]).dtype in (np.float32, np.float16, np.float64):
                            b = np.asarray(inits[other[0]], np.float32).reshape(-1)
                            if b.size == w.shape[0] or (n.op == "MatMulInteger" and b.size == w.shape[1]):
                                out[scope + ".bias"] = (b, f"bias added behind {n.op} (Add operand named by the node scope)", None)
                        break
                    if u.op not in ("Cast", "Mul", "Reshape", "Identity"):
                        break
                    cur, hops = u.outputs[0], hops + 1
        elif n.op == "MatMul" and scope and len(n.inputs) > 1 and n.inputs[1] in out and n.inputs[1] not in _named(out):
            out[scope] = (out[n.inputs[1]][0].T, "MatMul operand [in, out] transposed to [out, in]", None)
    return out


def _strip(name: str) -> str:
    for suf in ("_Q4", "_quantized", "_q4", "_int8"):
        if name.endswith(suf):
            return name[: -len(suf)]
    return name


def _named(out: dict) -> set:
    return {k for k in out if k.endswith((".weight", ".bias"))}
