"""Convert a FastConformer-CTC checkpoint into the flat weight file libqverse.so loads
(``qv_config.weights_path`` / ``QVERSE_WEIGHTS``).

    python tools/convert_weights.py --nemo stt_ar_fastconformer_hybrid_large_pcd.nemo --out fc.qvw
    python tools/convert_weights.py --state-dict model_weights.ckpt --out fc.qvw
    python tools/convert_weights.py --random 20260630 --out random.qvw      # seeded synthetic

File format ("QVWT0001"): u32 tensor count, then per tensor {u32 name_len, name, u32 numel,
float32 data}; names are the NeMo state-dict keys of the CTC branch (encoder.*,
ctc_decoder.decoder_layers.0.*), the same the engine's seeded init uses
(csrc/qv_model.hip::weight_shapes, read through the library's host-only qv_weight_spec).  A ``.nemo`` file is a tar archive holding
``model_weights.ckpt`` (a torch state dict), readable without NeMo.

    python tools/convert_weights.py --onnx fastconformer_full_mixed.onnx --list          # what the file holds
    python tools/convert_weights.py --onnx fastconformer_full_mixed.onnx --out fc.qvw [--map names.json]

``--onnx`` reads the reference's own weight format without onnx/onnxruntime (tools/onnx_reader.py):
MatMulNBits int4 blocks, DequantizeLinear'ed and ConvInteger/MatMulInteger int8 tensors are
dequantised to float32 and the file is MARKED pre-quantised (``qv.prequantised`` plus one
``<key>#int8_scale`` per int8 Conv weight): the engine then never re-quantises -- with
``precision=2`` (the reference's onnxruntime arithmetic) the Conv weights go back onto the file's
own integers with the file's own scale, and the Linear weights run as the dequantised values
(MatMulNBits zero points included); ``precision=0/1`` run the same values in fp16.  Tensors are matched to the NeMo state-dict keys by
initializer name, else by the scope of the node that consumes them
("/encoder/layers.0/feed_forward1/linear1/MatMul" -> encoder.layers.0.feed_forward1.linear1.weight);
``--map`` ({"nemo key": "onnx key"}) overrides, ``--list`` prints every candidate with its shape.
A BatchNorm that the exporter folded into the depthwise convolution is written as the identity.
The real file is absent from the build container: the reader is tested on synthetic models only
(tests/test_onnx_reader.py), and the name matching reports anything it cannot place.
"""

from __future__ import annotations

import argparse
import io
import struct
import sys
import tarfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


PREQUANT_KEY = "qv.prequantised"       # marker tensor: the file's weights already sit on their quantisation grids
SCALE_SUFFIX = "#int8_scale"           # "<NeMo key>#int8_scale": the per-tensor scale of an int8 Conv weight, verbatim


def write_qvw(path, tensors: dict, extra: dict | None = None):
    tensors = dict(tensors, **(extra or {}))
    with open(path, "wb") as f:
        f.write(b"QVWT0001")
        f.write(struct.pack("<I", len(tensors)))
        for name, arr in tensors.items():
            a = np.ascontiguousarray(np.asarray(arr, dtype=np.float32)).reshape(-1)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)))
            f.write(nb)
            f.write(struct.pack("<I", a.size))
            f.write(a.tobytes())


def load_state_dict(args):
    import torch

    if args.nemo:
        with tarfile.open(args.nemo) as tar:
            member = next(m for m in tar.getmembers() if m.name.endswith("model_weights.ckpt"))
            buf = io.BytesIO(tar.extractfile(member).read())
        return torch.load(buf, map_location="cpu", weights_only=True)
    return torch.load(args.state_dict, map_location="cpu", weights_only=True)


def _lib():
    """libqverse.so for its host-only weight-spec entry points (loads without a GPU)."""
    import ctypes as C

    path = ROOT / "offline-tarteel_amd" / "libqverse.so"
    if not path.exists():
        raise SystemExit(f"{path} not found: build it with `python offline-tarteel_amd/build.py`")
    lib = C.CDLL(str(path))
    lib.qv_weight_spec.argtypes = [C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.qv_weight_random.argtypes = [C.c_uint64, C.c_int32, C.c_void_p, C.c_int64]
    return lib


def weight_shapes(lib) -> dict:
    """{NeMo key: shape} in file order, as the engine defines them (qv_weight_spec)."""
    import ctypes as C

    out = {}
    for i in range(lib.qv_weight_count()):
        name = C.create_string_buffer(256)
        dims, nd = (C.c_int32 * 4)(), C.c_int32()
        if lib.qv_weight_spec(i, name, 256, dims, C.byref(nd)):
            raise SystemExit(f"qv_weight_spec({i}) failed")
        out[name.value.decode()] = tuple(dims[: nd.value])
    return out


def random_weights(lib, shapes: dict, seed: int) -> dict:
    out = {}
    for i, (name, shape) in enumerate(shapes.items()):
        a = np.empty(int(np.prod(shape)), np.float32)
        if lib.qv_weight_random(seed, i, a.ctypes.data, a.size):
            raise SystemExit(f"qv_weight_random({name}) failed")
        out[name] = a.reshape(shape)
    return out


def _fit(arr: np.ndarray, shape):
    """arr as `shape` if it is that tensor up to singleton dimensions, else None."""
    if tuple(arr.shape) == tuple(shape):
        return arr
    squeeze = lambda s: tuple(d for d in s if d != 1)  # noqa: E731
    if arr.size == int(np.prod(shape)) and squeeze(arr.shape) == squeeze(shape):
        return arr.reshape(shape)
    return None


def onnx_state_dict(path, shapes: dict, name_map: dict | None = None, verbose: bool = True, with_meta: bool = False):
    """NeMo-keyed float32 tensors from an ONNX file (see the module docstring).  with_meta: also
    {NeMo key: onnx_reader meta} for the tensors that came out of a quantised storage form."""
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import onnx_reader as O

    nodes, inits = O.read_model(path)
    fw = O.float_weights(nodes, inits)
    name_map = name_map or {}
    out, missing, meta = {}, [], {}
    for name, shape in shapes.items():
        cands = []
        if name in name_map:
            cands.append(name_map[name])
        cands.append(name)
        module = name.rsplit(".", 1)[0]
        if name.endswith((".weight", ".bias")):   # operands named by node scope, possibly with a different prefix
            tail = [k for k in fw if k.endswith(name.rsplit(".", 1)[1]) and k != name and
                    (k.endswith("." + name) or name.endswith("." + k)) and _fit(fw[k][0], shape) is not None]
            if len(tail) == 1:
                cands.append(tail[0])
        if name.endswith(".weight"):
            cands.append(module)
            # scopes may lack or carry extra leading components ("layers.0..." / "model.encoder.layers.0...")
            tail = [k for k in fw if not k.endswith((".weight", ".bias")) and
                    (k.endswith("." + module) or module.endswith("." + k)) and _fit(fw[k][0], shape) is not None]
            if len(tail) == 1:
                cands.append(tail[0])
        got = None
        for c in cands:
            if c in fw:
                got = _fit(fw[c][0], shape)
                if got is not None:
                    if fw[c][2] is not None:
                        meta[name] = fw[c][2]
                    break
        if got is None:
            missing.append(name)
        else:
            out[name] = got
    # BatchNorm folded into the depthwise convolution by the exporter: identity statistics
    for name in list(missing):
        if ".conv.batch_norm." in name:
            kind = name.rsplit(".", 1)[1]
            out[name] = np.full(shapes[name], {"weight": 1.0, "bias": 0.0, "running_mean": 0.0,
                                               "running_var": 1.0 - 1e-5}[kind], np.float32)
            missing.remove(name)
            if verbose and kind == "weight":
                print(f"note: {name.rsplit('.', 1)[0]} not in the graph (folded by the exporter): identity")
    if missing:
        raise SystemExit("cannot place these tensors (use --list and --map):\n  " + "\n  ".join(missing[:40]) +
                         (f"\n  ... and {len(missing) - 40} more" if len(missing) > 40 else ""))
    return (out, meta) if with_meta else out


def prequantised_extras(meta: dict) -> dict:
    """The extra tensors that tell the engine the file's weights are ALREADY quantised (precision 1 / 2 then never
    re-quantise them: Linear weights run as the dequantised values -- MatMulNBits blocks may carry zero points the
    engine's own symmetric int4 packing cannot express --, int8 Conv weights are put back on their integers with the
    file's scale instead of a re-derived max|w| / 127)."""
    extra = {PREQUANT_KEY: np.ones(1, np.float32)}
    for name, m in meta.items():
        if m.get("kind") == "int8":
            if m.get("op", "ConvInteger") != "ConvInteger":
                continue          # MatMulInteger Linear weights: the engine holds no int8 Linear path and reads no scale for them
            if m.get("per_channel") or int(m.get("zero_point", 0)) != 0:
                raise SystemExit(f"{name}: int8 Conv weight with " + ("per-channel scales" if m.get("per_channel") else
                                 f"zero point {m['zero_point']}") + " -- the engine's ConvInteger path holds ONE symmetric scale "
                                 "per weight tensor (onnxruntime quantize_dynamic's default); re-export with per_channel=False / "
                                 "symmetric weights, or convert the float model instead")
            extra[name + SCALE_SUFFIX] = np.array([m["scale"]], np.float32)
        elif m.get("kind") == "int4" and int(m.get("block", 0)) == 128 and "scales" in m:
            # the file's own MatMulNBits grid: the engine's W4A16 GEMM runs the file's integers with the file's zero points
            # (scale rounded to f16); other block sizes have no grid entry and run as the dequantised f16 values
            extra[name + "#int4_scale"] = np.asarray(m["scales"], np.float32).reshape(-1)
            extra[name + "#int4_zp"] = np.asarray(m["zero_points"], np.float32).reshape(-1)
    return extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nemo")
    ap.add_argument("--state-dict")
    ap.add_argument("--onnx")
    ap.add_argument("--map", help="JSON {nemo key: onnx key} overrides for --onnx")
    ap.add_argument("--list", action="store_true", help="with --onnx: print the weight candidates and exit")
    ap.add_argument("--random", type=int)
    ap.add_argument("--out")
    args = ap.parse_args()
    lib = _lib()
    shapes = weight_shapes(lib)
    if args.onnx and args.list:
        sys.path.insert(0, str(Path(__file__).resolve().parent))
        import onnx_reader as O

        nodes, inits = O.read_model(args.onnx)
        for k, (a, how, meta) in sorted(O.float_weights(nodes, inits).items()):
            q = "" if meta is None else ("  int4 blocks" if meta["kind"] == "int4" else f"  int8 scale {meta['scale']:.9g} zp {meta['zero_point']}")
            print(f"{k}  {tuple(a.shape)}  [{how}]{q}")
        ops = {}
        for n in nodes:
            ops[n.op] = ops.get(n.op, 0) + 1
        print("node types:", dict(sorted(ops.items())))
        return
    if not args.out:
        raise SystemExit("--out is required")
    if args.onnx:
        import json

        sd_np, meta = onnx_state_dict(args.onnx, shapes, json.loads(Path(args.map).read_text()) if args.map else None,
                                      with_meta=True)
        extra = prequantised_extras(meta) if meta else {}
        write_qvw(args.out, {k: sd_np[k] for k in shapes}, extra)
        n4 = sum(1 for m in meta.values() if m["kind"] == "int4")
        print(f"wrote {args.out}: {len(shapes)} tensors from {args.onnx}" +
              (f" ({n4} int4 MatMulNBits, {len(extra) - 1} int8 tensors with their scales; marked pre-quantised)" if extra else ""))
        return
    if args.random is not None:
        write_qvw(args.out, random_weights(lib, shapes, args.random))
        print(f"wrote {args.out}: {len(shapes)} seeded synthetic tensors (seed {args.random})")
        return
    sd = load_state_dict(args)
    if "state_dict" in sd:
        sd = sd["state_dict"]
    out = {}
    for name, shape in shapes.items():
        if name not in sd:
            raise SystemExit(f"checkpoint lacks {name}")
        t = sd[name].float().numpy()
        if tuple(t.shape) != tuple(shape):
            raise SystemExit(f"{name}: shape {tuple(t.shape)} != expected {tuple(shape)}")
        out[name] = t
    write_qvw(args.out, out)
    print(f"wrote {args.out}: {len(out)} tensors, {sum(v.size for v in out.values()) / 1e6:.1f} M parameters")


if __name__ == "__main__":
    main()
