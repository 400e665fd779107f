#!/usr/bin/env python
"""Does it matter whether the engine is created before or after the process's first other device work?

    python tools/init_order_probe.py engine_first | torch_first | hip_memcpy_first [--contexts 4]

Prints one JSON line: ms per batch of 64 x 10 s clips, `contexts` batches in flight, no results fetched (tools/sweep.py's
loop).  Until round 5 `engine_first` read 4.7-4.8 ms against 3.5 for the other two: the legacy default stream -- the
stream the batches are ordered behind -- got its hardware queue only after the engine's streams and shared one with a
context stream (qv_create now runs one kernel on it first; profiles/archive/r05_r_init_order.log,
tests/test_gpu_bench.py::test_engine_created_before_any_other_device_work_runs_at_full_speed)."""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("order", choices=("engine_first", "torch_first", "hip_memcpy_first"))
    ap.add_argument("--contexts", type=int, default=4)
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()

    import offline_tarteel_amd  # noqa: F401  (first: sets GPU_MAX_HW_QUEUES before HIP initialises)
    import torch
    from offline_tarteel_amd.engine import Engine
    from synth import synth_audio

    B, n = 64, 160000
    host = torch.from_numpy(synth_audio(B, n, seed=20260630))
    audio = None
    if args.order == "torch_first":
        audio = host.cuda().contiguous()
    elif args.order == "hip_memcpy_first":
        hip = C.CDLL("libamdhip64.so")
        p, buf = C.c_void_p(), (C.c_char * (1 << 20))()
        hip.hipSetDevice(0)
        hip.hipMalloc(C.byref(p), C.c_size_t(1 << 20))
        hip.hipMemcpy(p, buf, C.c_size_t(1 << 20), 1)
    eng = Engine(device=0, with_model=True, seed=20260630, max_batch=B, max_samples=n, contexts=args.contexts)
    if audio is None:
        audio = host.cuda().contiguous()
    lens = [n] * B
    for _ in range(8):
        eng.predict_batch_async(audio, lens)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.predict_batch_async(audio, lens)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    print(json.dumps({"order": args.order, "contexts": eng.contexts, "ms_per_batch": round(ms, 3),
                      "utterances_per_s": round(B / ms * 1e3, 1)}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
