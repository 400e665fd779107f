"""Host-side mirror of the reference's plugin surface for this path
(experiments/c2c-direct-mixed/run.py and experiments/c2c-direct-mixed-tta/run.py):
``predict(audio_path) -> dict``, ``transcribe(audio_path) -> str``, ``model_size() -> int``,
plus the batched entry the throughput metric needs (``predict_batch``).

Same dict keys (surah, ayah, ayah_end, score, transcript, source; _empty on no match), same env
knobs (CTC_DIRECT_*, C2C_DIRECT_MIXED_PROFILE), same error behaviour (missing model ->
FileNotFoundError; any exception propagates to the runner, which records an empty prediction).

Model weights: ``QVERSE_WEIGHTS`` names the flat weight file produced by
tools/convert_weights.py from the reference's checkpoint.  Without it ``predict`` raises
FileNotFoundError like the reference does for a missing ONNX -- unless
``QVERSE_RANDOM_WEIGHTS=1`` explicitly asks for the seeded synthetic weights (throughput runs).
"""

from __future__ import annotations

import os
import time
from collections import Counter
from pathlib import Path

import numpy as np

from .audio import load_audio

_PROFILE = os.getenv("C2C_DIRECT_MIXED_PROFILE", "") not in ("", "0", "false", "False")
CONFIDENCE_SKIP_THRESHOLD = 0.5  # c2c-direct-mixed-tta/run.py:57
MAX_SAMPLES = int(os.getenv("QVERSE_MAX_SAMPLES", str(16000 * 60)))
MAX_BATCH = int(os.getenv("QVERSE_MAX_BATCH", "16"))

# QVERSE_PRECISION: "fp16" (default), "mixed" (the engine's own W4A16 / W8A16 kernels) or "ort" -- the arithmetic
# onnxruntime runs on the reference's fastconformer_full_mixed.onnx (include/qverse.h QV_PREC_ORT_MIXED); with a weight
# file converted from that ONNX (marked pre-quantised) "ort" computes on the file's own integers
_PRECISIONS = {"fp16": 0, "mixed": 1, "ort": 2}
_engine = None
_last_raw: list[dict] = []   # engine-level dicts of the most recent predict_arrays() call (profile line)


def weights_path() -> Path | None:
    p = os.getenv("QVERSE_WEIGHTS")
    return Path(p) if p else None


def _ensure_engine():
    global _engine
    if _engine is not None:
        return _engine
    from .engine import Engine

    wp = weights_path()
    if wp is None and os.getenv("QVERSE_RANDOM_WEIGHTS", "") not in ("1", "true", "True"):
        raise FileNotFoundError(
            "No model weights: set QVERSE_WEIGHTS to the flat file written by tools/convert_weights.py "
            "(from fastconformer_full_mixed.onnx / the .nemo checkpoint), or QVERSE_RANDOM_WEIGHTS=1 for "
            "seeded synthetic weights.")
    if wp is not None and not wp.exists():
        raise FileNotFoundError(f"No weight file at {wp}. Run `python tools/convert_weights.py` first.")
    device = int(os.getenv("LOCAL_RANK", "0"))
    prec_name = os.getenv("QVERSE_PRECISION", "fp16")
    if prec_name not in _PRECISIONS:
        raise ValueError(f"QVERSE_PRECISION={prec_name!r}: expected one of {sorted(_PRECISIONS)}")
    print(f"[c2c-direct-mixed/qverse] loading {'synthetic weights' if wp is None else wp.name} on cuda:{device} ({prec_name})...")
    _engine = Engine(device=device, with_model=True, weights_path=str(wp) if wp else None, precision=_PRECISIONS[prec_name],
                     max_batch=MAX_BATCH, max_samples=MAX_SAMPLES)
    if _PROFILE:
        _engine.profile_stages(True)
    return _engine


def _empty(transcript: str = "") -> dict:
    return {"surah": 0, "ayah": 0, "ayah_end": None, "score": 0.0, "transcript": transcript, "candidates": []}


def _to_dict(r: dict, round_score: bool) -> dict:
    if not r["surah"]:
        return _empty(r.get("transcript", ""))
    score = r["score"]
    return {
        "surah": r["surah"], "ayah": r["ayah"], "ayah_end": r["ayah_end"] or r["ayah"],
        "score": round(score, 4) if round_score else score,
        "transcript": r.get("transcript", ""), "source": r["source"],
    }


def predict_arrays(arrays, round_score: bool = True) -> list[dict]:
    """audio arrays (float32, 16 kHz) -> predict()-shaped dicts, one engine call per <= MAX_BATCH."""
    import torch

    eng = _ensure_engine()
    out = []
    for s in range(0, len(arrays), eng.max_batch):
        chunk = arrays[s: s + eng.max_batch]
        lens = [len(a) for a in chunk]
        buf = np.zeros((len(chunk), max(lens)), dtype=np.float32)
        for i, a in enumerate(chunk):
            buf[i, : len(a)] = a
        dev = torch.from_numpy(buf).cuda(eng.device)
        raw = eng.predict_batch(dev, lens)
        _last_raw[:] = raw
        out.extend(_to_dict(r, round_score) for r in raw)
    return out


def predict_device(dev, lens, round_score: bool = True) -> list[dict]:
    """a zero-padded float32 cuda matrix of 16 kHz clips -> predict()-shaped dicts, one engine call per <= MAX_BATCH rows"""
    eng = _ensure_engine()
    out = []
    for s in range(0, len(lens), eng.max_batch):
        chunk = lens[s: s + eng.max_batch]
        raw = eng.predict_batch(dev[s: s + len(chunk), : max(chunk)].contiguous(), chunk)
        _last_raw[:] = raw
        out.extend(_to_dict(r, round_score) for r in raw)
    return out


def predict_batch(audio_paths) -> list[dict]:
    """files -> dicts.  The ingest (mix-down, resampling to 16 kHz) runs on the GPU (audio.load_audio_device: the same float32
    arithmetic as the host load_audio, bit for bit); QVERSE_INGEST=host keeps it on the host."""
    if os.getenv("QVERSE_INGEST", "device") == "host":
        return predict_arrays([load_audio(p) for p in audio_paths])
    from .audio import load_audio_device

    dev, lens = load_audio_device(list(audio_paths), _ensure_engine())
    return predict_device(dev, lens)


def predict(audio_path: str) -> dict:
    t0 = time.perf_counter()
    audio = load_audio(audio_path)
    t1 = time.perf_counter()
    res = predict_arrays([audio])[0]
    if _PROFILE:
        # the reference's line (mixed/run.py:76-81,117-124): forward= decode= build= rerank= total=
        # candidates= use_ctc= source= -- stage times from HIP events around the device stages of this
        # file's batch of one; `forward` includes the audio load like the reference's _ctc_logprobs does
        t2 = time.perf_counter()
        st = _ensure_engine().stage_times()
        raw = _last_raw[0] if _last_raw else {}
        head = f"[c2c-direct-mixed profile] audio={Path(audio_path).name} "
        fwd = (t1 - t0) + st["forward"]
        if not res.get("transcript", "").strip():
            print(head + f"forward={fwd:.3f}s decode={st['decode']:.3f}s total={t2 - t0:.3f}s empty=1")
        elif not res["surah"]:
            print(head + f"forward={fwd:.3f}s decode={st['decode']:.3f}s build={st['build']:.3f}s "
                         f"total={t2 - t0:.3f}s no_candidates=1")
        else:
            use_ctc = int(bool(raw.get("use_ctc")))
            print(head + f"forward={fwd:.3f}s decode={st['decode']:.3f}s build={st['build']:.3f}s "
                         f"rerank={st['rerank'] if use_ctc else 0.0:.3f}s total={t2 - t0:.3f}s "
                         f"candidates={raw.get('n_candidates', 0)} use_ctc={use_ctc} source={res.get('source')}")
    return res


def transcribe(audio_path: str) -> str:
    import torch

    eng = _ensure_engine()
    audio = load_audio(audio_path)
    dev = torch.from_numpy(audio[None, :]).cuda(eng.device)
    lp, T = eng.forward(dev, [len(audio)])
    ids = lp[0, : T[0]].argmax(-1).cpu().numpy().tolist()
    dedup, prev = [], -1
    for i in ids:
        if i != prev and i != 1024:
            dedup.append(i)
        prev = i
    return eng.transcript_of(dedup)


def model_size() -> int:
    wp = weights_path()
    return wp.stat().st_size if wp is not None and wp.exists() else 0


# ------------------------------------------------------------------ TTA variant --------
def _tta_combine(p09: dict, anchor: dict, p11: dict) -> dict:
    """c2c-direct-mixed-tta/run.py:133-149: majority over (surah, ayah) of [0.9x, anchor, 1.1x],
    else the highest score."""
    preds = [p09, anchor, p11]
    keys = [(p["surah"], p["ayah"]) for p in preds]
    top, n = Counter(keys).most_common(1)[0]
    if n >= 2:
        for p in preds:
            if (p["surah"], p["ayah"]) == top:
                p["tta"] = "majority"
                p["tta_preds"] = keys
                return p
    best = max(preds, key=lambda p: p["score"])
    best["tta"] = "score_pick"
    best["tta_preds"] = keys
    best["tta_scores"] = [p["score"] for p in preds]
    return best


def tta_start(eng, audio, lengths, want_text: bool = True, anchor_ctx: int | None = None) -> dict:
    """First half of tta_device_batch: the anchor results (run here, or fetched from context `anchor_ctx` when the
    caller already launched the anchor pass with predict_batch_async), the 0.5 gate, and the 0.9x / 1.1x copies
    of the gated clips launched as further batches.  With one context those batches are finished here; with more
    they stay in flight until tta_finish() -- a serving loop can launch the next batch's anchor pass in between.

    Round 6: the copies of ALL gated clips of one speed are made by ONE launch (Engine.speed_perturb_rows ->
    qv_upfirdn_batch, straight into the engine's zero-padded input layout) and travel as one batch per speed -- rows of
    similar length, and only the slowed copies can exceed 384 frames -- instead of one launch + one row copy per clip and
    batches that interleave the two speeds."""
    if anchor_ctx is None:
        raw = eng.predict_batch(audio, lengths, want_text=want_text)
    else:
        raw = eng.fetch_results(anchor_ctx, len(lengths), eng.frames_for(max(lengths)), want_text=want_text)
    anchors = [_to_dict(r, False) for r in raw]
    hard = [i for i, a in enumerate(anchors) if a["score"] < CONFIDENCE_SKIP_THRESHOLD]
    st = {"anchors": anchors, "out": list(anchors), "tickets": [], "want_text": want_text, "parts": {}}
    for s0 in range(0, len(hard), eng.max_batch):
        idx = hard[s0: s0 + eng.max_batch]
        lens_in = [lengths[i] for i in idx]
        for factor in (0.9, 1.1):
            rows, lens = eng.speed_perturb_rows(audio, lens_in, factor, src_rows=idx)
            if eng.contexts > 1:
                # the perturbed batches do not depend on one another: keep them in flight (one context stays free for
                # the caller's next anchor pass when there are more than two)
                if len(st["tickets"]) >= max(1, eng.contexts - (1 if eng.contexts > 2 else 0)):
                    _tta_join(eng, st, st["tickets"].pop(0))
                st["tickets"].append((eng.predict_batch_async(rows, lens), idx, rows, eng.frames_for(max(lens)), factor))
                continue
            res = [_to_dict(r, False) for r in eng.predict_batch(rows, lens, want_text=want_text)]
            _tta_part(st, idx, factor, res)
    return st


def _tta_part(st: dict, idx, factor: float, res: list[dict]):
    """one speed's results of the clips `idx`; a clip is decided once both of its copies are in"""
    for k, i in enumerate(idx):
        part = st["parts"].setdefault(i, {})
        part[factor] = res[k]
        if len(part) == 2:
            st["out"][i] = _tta_combine(part[0.9], st["anchors"][i], part[1.1])
            del st["parts"][i]


def _tta_join(eng, st: dict, ticket):
    ctx, idx, rows, t_max, factor = ticket   # (rows: the batch's input, kept alive until it has run)
    res = [_to_dict(r, False) for r in eng.fetch_results(ctx, rows.shape[0], t_max, want_text=st["want_text"])]
    _tta_part(st, idx, factor, res)


def tta_finish(eng, st: dict) -> list[dict]:
    """Second half: join the perturbed batches still in flight and apply the majority / best-score rule."""
    while st["tickets"]:
        _tta_join(eng, st, st["tickets"].pop(0))
    return st["out"]


def tta_device_batch(eng, audio, lengths, want_text: bool = True) -> list[dict]:
    """c2c-direct-mixed-tta/run.py:117-149 for one batch already in HBM (float32 cuda [B, N], zero
    padded): anchor pass over every clip, gate 0.5 on the UNROUNDED score, then the 0.9x / 1.1x copies of
    the gated clips -- made on the GPU (qv_upfirdn, bit-identical to the reference's
    scipy.signal.resample_poly call) -- as further engine batches (the reference runs them as two
    threads on one session), and the majority / best-score rule."""
    return tta_finish(eng, tta_start(eng, audio, lengths, want_text))


def predict_tta_arrays(arrays) -> list[dict]:
    """tta_device_batch for a list of host clips (one engine batch per <= MAX_BATCH clips)."""
    import torch

    eng = _ensure_engine()
    out = []
    for s0 in range(0, len(arrays), eng.max_batch):
        chunk = [np.ascontiguousarray(a, dtype=np.float32) for a in arrays[s0: s0 + eng.max_batch]]
        lens = [len(a) for a in chunk]
        buf = np.zeros((len(chunk), max(lens)), dtype=np.float32)
        for i, a in enumerate(chunk):
            buf[i, : len(a)] = a
        out.extend(tta_device_batch(eng, torch.from_numpy(buf).cuda(eng.device), lens))
    return out


def predict_tta(audio_path: str) -> dict:
    return predict_tta_arrays([load_audio(audio_path)])[0]


def predict_tta_batch(audio_paths) -> list[dict]:
    return predict_tta_arrays([load_audio(p) for p in audio_paths])
