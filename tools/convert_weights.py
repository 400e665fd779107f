"""Convert a FastConformer-CTC checkpoint into the flat weight file libqverse.so loads
(``qv_config.weights_path`` / ``QVERSE_WEIGHTS``).

    python tools/convert_weights.py --nemo stt_ar_fastconformer_hybrid_large_pcd.nemo --out fc.qvw
    python tools/convert_weights.py --state-dict model_weights.ckpt --out fc.qvw
    python tools/convert_weights.py --random 20260630 --out random.qvw      # seeded synthetic

File format ("QVWT0001"): u32 tensor count, then per tensor {u32 name_len, name, u32 numel,
float32 data}; names are the NeMo state-dict keys of the CTC branch (encoder.*,
ctc_decoder.decoder_layers.0.*), the same the engine's seeded init uses
(csrc/qv_model.hip::weight_shapes).  A ``.nemo`` file is a tar archive holding
``model_weights.ckpt`` (a torch state dict), readable without NeMo.

The mixed int4/int8 ONNX (fastconformer_full_mixed.onnx, MatMulNBits + dynamic-int8 Conv) is
NOT handled yet: it needs a protobuf reader and block dequantisation (next round; the fp16 path
then reproduces the *dequantised* weights, while ORT's dynamic activation quantisation for Conv
stays a documented numerical difference).
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


def write_qvw(path, tensors: dict):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nemo")
    ap.add_argument("--state-dict")
    ap.add_argument("--random", type=int)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    from oracle import fastconformer_ref as R  # shapes + seeded init only (build-time tool)

    shapes = R.weight_shapes()
    if args.random is not None:
        sd = R.random_weights(args.random)
    else:
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
