"""ctypes binding of libqverse.so (include/qverse.h) + the host-side glue the plugin needs.

PyTorch-ROCm is used for device memory and streams only: tensors are handed to the C ABI as
raw device pointers.  There is NO CPU fallback: if the HIP library is missing or no GPU is
visible, construction raises.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

from . import LIB_PATH, TABLES_PATH
from .normalizer import normalize_arabic
from .tables import Tables

QV_SOURCE = {0: None, 1: "text", 2: "ctc"}
QV_MAX_TRANSCRIPT = 1024   # include/qverse.h: characters the device matchers hold per text


def front_window(text: str, limit: int = QV_MAX_TRANSCRIPT) -> str:
    """The longest whole-word prefix of `text` that fits the device's matching window.  The reference's
    single-text matchers (QuranDB.match_verse behind run_on_full_transcript, VerseTracker._find_best_match)
    take texts of any length; both match -- and then peel -- the FRONT of the text, so a longer text is
    matched on its first `limit` characters instead of being refused (documented difference)."""
    if len(text) <= limit:
        return text
    cut = text.rfind(" ", 0, limit + 1)
    return text[:cut] if cut > 0 else text[:limit]
FLAG_EMPTY, FLAG_TRUNC, FLAG_USED_CTC, FLAG_CAND_OVERFLOW = 1, 2, 4, 8


class QvError(RuntimeError):
    pass


class QvConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32),
        ("tables_path", C.c_char_p), ("weights_path", C.c_char_p),
        ("random_weights_seed", C.c_uint64),
        ("with_model", C.c_int32), ("precision", C.c_int32),
        ("max_batch", C.c_int32), ("max_samples", C.c_int32),
        ("top_text", C.c_int32), ("top_span_refs", C.c_int32), ("max_span", C.c_int32),
        ("threshold", C.c_double), ("text_weight", C.c_double), ("span_penalty", C.c_double),
        ("skip_unused_passes", C.c_int32), ("n_contexts", C.c_int32),
    ]


class QvResult(C.Structure):
    _fields_ = [
        ("surah", C.c_int32), ("ayah", C.c_int32), ("ayah_end", C.c_int32), ("source", C.c_int32),
        ("score", C.c_double), ("base_score", C.c_double), ("ctc_norm_loss", C.c_float),
        ("n_tokens", C.c_int32), ("n_chars", C.c_int32), ("n_candidates", C.c_int32),
        ("flags", C.c_int32), ("t_frames", C.c_int32),
    ]


class QvTrackMatch(C.Structure):
    """include/qverse.h: qv_track_match"""
    _fields_ = [("verse", C.c_int32), ("surah", C.c_int32), ("ayah", C.c_int32), ("variant", C.c_int32),
                ("n_words", C.c_int32), ("reserved", C.c_int32), ("score", C.c_double)]


RESULT_DTYPE = np.dtype([
    ("surah", "<i4"), ("ayah", "<i4"), ("ayah_end", "<i4"), ("source", "<i4"),
    ("score", "<f8"), ("base_score", "<f8"), ("ctc_norm_loss", "<f4"),
    ("n_tokens", "<i4"), ("n_chars", "<i4"), ("n_candidates", "<i4"),
    ("flags", "<i4"), ("t_frames", "<i4"),
], align=True)
assert RESULT_DTYPE.itemsize == C.sizeof(QvResult)

_lib = None


def load_library(path: Path | str | None = None) -> C.CDLL:
    """Load libqverse.so; fail loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    p = Path(path or os.environ.get("QVERSE_LIB", LIB_PATH))
    if not p.exists():
        raise QvError(f"{p} not found: build it with `python offline-tarteel_amd/build.py` "
                      "(hipcc, gfx950). There is no CPU fallback for this path.")
    lib = C.CDLL(str(p))
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.qv_config_default.argtypes = [C.POINTER(QvConfig)]
    lib.qv_config_default.restype = None
    lib.qv_create.argtypes = [C.POINTER(QvConfig), C.POINTER(vp)]
    lib.qv_destroy.argtypes = [vp]
    lib.qv_destroy.restype = None
    lib.qv_last_error.argtypes = [vp]
    lib.qv_last_error.restype = C.c_char_p
    lib.qv_build_info.restype = C.c_char_p
    lib.qv_frames_for_samples.argtypes = [i64]
    lib.qv_frames_for_samples.restype = i32
    lib.qv_forward.argtypes = [vp, vp, vp, i32, i64, vp, i32, vp, vp]
    lib.qv_decode_retrieve_rerank.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    lib.qv_decode_retrieve_rerank_async.argtypes = [vp, vp, vp, i32, i32, vp]
    lib.qv_fetch_results.argtypes = [vp, i32, i32, vp, vp, vp]
    lib.qv_predict_batch.argtypes = [vp, vp, vp, i32, i64, vp, vp, vp]
    lib.qv_predict_batch_async.argtypes = [vp, vp, vp, i32, i64, vp]
    lib.qv_predict_batch_async_ctx.argtypes = [vp, vp, vp, i32, i64, vp, vp]
    lib.qv_packed_results_dev.argtypes = [vp]
    lib.qv_packed_results_dev.restype = vp
    lib.qv_upfirdn.argtypes = [vp, vp, i64, vp, i32, i32, i32, i64, i64, vp, vp]
    lib.qv_upfirdn_batch.argtypes = [vp, vp, i64, vp, vp, i32, vp, i32, i32, i32, i64, vp, i64, vp]
    lib.qv_mixdown_batch.argtypes = [vp, vp, i64, vp, i32, i32, vp, i64, vp]
    lib.qv_context_count.argtypes = [vp]
    lib.qv_probe_concurrent_streams.argtypes = []
    lib.qv_last_context.argtypes = [vp]
    lib.qv_wait_ctx.argtypes = [vp, i32]
    lib.qv_packed_results_ctx.argtypes = [vp, i32, vp]
    lib.qv_packed_results_ctx.restype = vp
    lib.qv_fetch_results_ctx.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.qv_tracker_match.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp]
    lib.qv_match_verse.argtypes = [vp, vp, i32, i32, vp, vp, i32, vp, vp, vp, vp]
    lib.qv_debug_retrieve.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.qv_debug_ctc_loss.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp]
    lib.qv_debug_forward_tap.argtypes = [vp, i32, i32, vp, vp]
    lib.qv_profile_gemm.argtypes = [vp, i32]
    lib.qv_profile_gemm_read.argtypes = [vp, vp, vp, vp]
    lib.qv_profile_replay_gemm.argtypes = [vp, i32, i32, vp, vp, vp]
    lib.qv_profile_replay_kernel.argtypes = [vp, i32, C.c_char_p, i32]
    lib.qv_debug_gemm_tiles.argtypes = [i32]
    lib.qv_debug_gemm_tile_height.argtypes = [i32]
    lib.qv_debug_forward_graph_failures.argtypes = [vp]
    lib.qv_debug_forward_graph_failures.restype = i64
    lib.qv_debug_attention_variant.argtypes = [i32]
    lib.qv_debug_kernel_variant.argtypes = [i32, i32]
    lib.qv_debug_forward_graph_stats.argtypes = [vp, vp, vp]
    lib.qv_weights_info.argtypes = [vp, C.c_char_p, i32]
    lib.qv_profile_inject_logprobs.argtypes = [vp, vp, i32, vp, i32]
    lib.qv_profile_stages.argtypes = [vp, i32]
    lib.qv_stage_times.argtypes = [vp, i32, vp]
    _lib = lib
    return lib


def exported_symbols() -> list[str]:
    """Every function include/qverse.h declares (used by the CPU load/export test)."""
    import re

    hdr = (Path(__file__).resolve().parent.parent / "include" / "qverse.h").read_text()
    return sorted(set(re.findall(r"\b(qv_[a-z_0-9]+)\s*\(", hdr)))


def env_knobs() -> dict:
    """CTC_DIRECT_* environment knobs with the reference's names and defaults
    (experiments/c2c-direct/run.py:62-74)."""
    return dict(
        top_text=int(os.getenv("CTC_DIRECT_TOP_TEXT", "100")),
        top_span_refs=int(os.getenv("CTC_DIRECT_TOP_SPAN_REFS", "80")),
        max_span=int(os.getenv("CTC_DIRECT_MAX_SPAN", "6")),
        threshold=float(os.getenv("CTC_DIRECT_THRESHOLD", "0.80")),
        text_weight=float(os.getenv("CTC_DIRECT_TEXT_WEIGHT", "0.0")),
        span_penalty=float(os.getenv("CTC_DIRECT_SPAN_PENALTY", "0.5")),
    )


class Engine:
    """One engine per process per GPU (weights + verse tables resident for its lifetime)."""

    def __init__(self, device: int = 0, with_model: bool = True, weights_path: str | None = None,
                 seed: int = 20260630, precision: int = 0, max_batch: int = 64, max_samples: int = 480000,
                 tables_path: str | None = None, skip_unused_passes: bool = True, contexts: int = 1, **knobs):
        import torch

        if not torch.cuda.is_available():
            raise QvError("no ROCm device visible: the qverse hot path runs on MI355X only (no CPU fallback)")
        self.torch = torch
        self.lib = load_library()
        self.device = device
        self.tables_path = str(tables_path or TABLES_PATH)
        if not Path(self.tables_path).exists():
            raise FileNotFoundError(self.tables_path)
        if weights_path is not None and not Path(weights_path).exists():
            raise FileNotFoundError(f"No weight file at {weights_path}")
        cfg = QvConfig()
        self.lib.qv_config_default(C.byref(cfg))
        cfg.device = device
        self._keep = (self.tables_path.encode(), weights_path.encode() if weights_path else None)
        cfg.tables_path, cfg.weights_path = self._keep
        cfg.random_weights_seed = seed
        cfg.with_model = int(with_model)
        cfg.precision = precision
        cfg.max_batch = max_batch
        cfg.max_samples = max_samples
        cfg.skip_unused_passes = int(skip_unused_passes)
        cfg.n_contexts = int(contexts)
        self.contexts = int(contexts)
        kn = env_knobs()
        kn.update(knobs)
        for k, v in kn.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        h = C.c_void_p()
        with torch.cuda.device(device):
            rc = self.lib.qv_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            msg = self.lib.qv_last_error(None).decode()
            if rc == 2:
                raise FileNotFoundError(msg)
            raise QvError(f"qv_create failed ({rc}): {msg}")
        self.h = h
        # the engine may run fewer batches in flight than asked for (qv_probe_concurrent_streams, include/qverse.h)
        self.contexts = int(self.lib.qv_context_count(h))
        self.max_batch = max_batch
        self.tables = Tables(self.tables_path)

    def close(self):
        if getattr(self, "h", None):
            self.lib.qv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- helpers ----
    def _check(self, rc: int, what: str):
        if rc != 0:
            raise QvError(f"{what} failed ({rc}): {self.lib.qv_last_error(self.h).decode()}")

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def frames_for(self, n_samples: int) -> int:
        return int(self.lib.qv_frames_for_samples(int(n_samples)))

    def _results(self, res: np.ndarray, greedy: np.ndarray | None) -> list[dict]:
        out = []
        for b in range(len(res)):
            r = res[b]
            d = {
                "surah": int(r["surah"]), "ayah": int(r["ayah"]),
                "ayah_end": int(r["ayah_end"]) if r["surah"] else None,
                "score": float(r["score"]), "source": QV_SOURCE[int(r["source"])],
                "base_score": float(r["base_score"]), "ctc_norm_loss": float(r["ctc_norm_loss"]),
                "n_candidates": int(r["n_candidates"]), "flags": int(r["flags"]),
                "t_frames": int(r["t_frames"]), "use_ctc": bool(r["flags"] & FLAG_USED_CTC),
            }
            if greedy is not None:
                ids = greedy[b, : int(r["n_tokens"])].tolist()
                d["greedy_ids"] = ids
                d["transcript"] = self.transcript_of(ids)
            out.append(d)
        return out

    def transcript_of(self, ids) -> str:
        """c2c-direct/run.py:201-204: ids_to_text(...).strip() then normalize_arabic."""
        if not ids:
            return ""
        return normalize_arabic(self.tables.ids_to_text(ids).strip())

    # ---------------------------------------------------------------- stages -----
    def forward(self, audio, lengths):
        """audio: float32 cuda tensor [B, N] (zero padded); lengths: int sequence.
        Returns (log_probs [B, Tmax, 1025] cuda float32, T list)."""
        torch = self.torch
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.is_contiguous()
        B, N = audio.shape
        ln = np.ascontiguousarray(np.asarray(lengths, dtype=np.int64))
        assert len(ln) == B and ln.max() <= N
        t_max = self.frames_for(int(ln.max()))
        lp = torch.empty((B, t_max, 1025), dtype=torch.float32, device=audio.device)
        t_out = np.zeros(B, dtype=np.int32)
        rc = self.lib.qv_forward(self.h, C.c_void_p(audio.data_ptr()), ln.ctypes.data_as(C.c_void_p), B, N,
                                 C.c_void_p(lp.data_ptr()), t_max, t_out.ctypes.data_as(C.c_void_p), self._stream())
        self._check(rc, "qv_forward")
        return lp, t_out.tolist()

    def decode_retrieve_rerank(self, log_probs, t_frames, want_text: bool = True) -> list[dict]:
        torch = self.torch
        assert log_probs.is_cuda and log_probs.dtype == torch.float32 and log_probs.is_contiguous()
        B, t_max, V = log_probs.shape
        assert V == 1025
        t = np.ascontiguousarray(np.asarray(t_frames, dtype=np.int32))
        res = np.zeros(B, dtype=RESULT_DTYPE)
        greedy = np.full((B, t_max), -1, dtype=np.int32) if want_text else None
        rc = self.lib.qv_decode_retrieve_rerank(
            self.h, C.c_void_p(log_probs.data_ptr()), t.ctypes.data_as(C.c_void_p), B, t_max,
            res.ctypes.data_as(C.c_void_p),
            greedy.ctypes.data_as(C.c_void_p) if greedy is not None else None, self._stream())
        self._check(rc, "qv_decode_retrieve_rerank")
        return self._results(res, greedy)

    def predict_batch(self, audio, lengths, want_text: bool = True) -> list[dict]:
        torch = self.torch
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.is_contiguous()
        B, N = audio.shape
        ln = np.ascontiguousarray(np.asarray(lengths, dtype=np.int64))
        t_max = self.frames_for(int(ln.max()))
        res = np.zeros(B, dtype=RESULT_DTYPE)
        greedy = np.full((B, t_max), -1, dtype=np.int32) if want_text else None
        rc = self.lib.qv_predict_batch(
            self.h, C.c_void_p(audio.data_ptr()), ln.ctypes.data_as(C.c_void_p), B, N,
            res.ctypes.data_as(C.c_void_p),
            greedy.ctypes.data_as(C.c_void_p) if greedy is not None else None, self._stream())
        self._check(rc, "qv_predict_batch")
        return self._results(res, greedy)

    def predict_batch_async(self, audio, lengths):
        B, N = audio.shape
        ln = np.ascontiguousarray(np.asarray(lengths, dtype=np.int64))
        ctx = C.c_int32(-1)
        rc = self.lib.qv_predict_batch_async_ctx(self.h, C.c_void_p(audio.data_ptr()), ln.ctypes.data_as(C.c_void_p), B, N,
                                                 self._stream(), C.byref(ctx))
        self._check(rc, "qv_predict_batch_async_ctx")
        return int(ctx.value)     # (the id comes back from the call itself: another thread cannot get in between)

    def fetch_results(self, ctx: int, batch: int, t_max: int, want_text: bool = False) -> list[dict]:
        """join context `ctx` (the value predict_batch_async returned) and copy its results out."""
        res = np.zeros(batch, dtype=RESULT_DTYPE)
        greedy = np.full((batch, t_max), -1, dtype=np.int32) if want_text else None
        rc = self.lib.qv_fetch_results_ctx(self.h, ctx, batch, t_max, res.ctypes.data_as(C.c_void_p),
                                           greedy.ctypes.data_as(C.c_void_p) if greedy is not None else None)
        self._check(rc, "qv_fetch_results_ctx")
        return self._results(res, greedy)

    def wait(self, ctx: int):
        """host-side join of context `ctx` (the value predict_batch_async returned): returns when its batch is done."""
        self._check(self.lib.qv_wait_ctx(self.h, int(ctx)), "qv_wait_ctx")

    def packed_results(self, batch: int, ctx: int | None = None):
        """int32 cuda tensor [batch, 4] = (surah, ayah, ayah_end, float-bits(score)) of the last
        async call -- or, with batches in flight, of context `ctx` (joined on the current stream).
        This is the payload of the per-batch all-gather."""
        torch = self.torch
        if ctx is None:
            ctx = int(self.lib.qv_last_context(self.h))
        ptr = self.lib.qv_packed_results_ctx(self.h, ctx, self._stream())
        if not ptr:
            raise QvError("qv_packed_results_ctx failed")
        out = torch.empty((batch, 4), dtype=torch.int32, device=f"cuda:{self.device}")
        # device-to-device copy out of the engine-owned buffer on the current stream
        import ctypes

        hip = _hip_runtime()
        rc = hip.hipMemcpyAsync(C.c_void_p(out.data_ptr()), C.c_void_p(ptr), batch * 16, 3, self._stream())
        if rc != 0:
            raise QvError(f"hipMemcpyAsync failed ({rc})")
        return out

    def resample_poly(self, x, up: int, down: int):
        """scipy.signal.resample_poly(x, up, down) for a float32 cuda vector, computed on the GPU
        and bit-identical to scipy's float32 result (a15: the TTA wrapper's 0.9x / 1.1x speed
        perturbation, c2c-direct-mixed-tta/run.py:60-71)."""
        from .audio import resample_plan

        torch = self.torch
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 1 and x.is_contiguous()
        n_in = int(x.numel())
        up, down, taps, m0, n_out = resample_plan(up, down, n_in)
        if up == down == 1:
            return x.clone()
        y = torch.empty(n_out, dtype=torch.float32, device=x.device)
        rc = self.lib.qv_upfirdn(self.h, C.c_void_p(x.data_ptr()), n_in, taps.ctypes.data_as(C.c_void_p), len(taps),
                                 up, down, m0, n_out, C.c_void_p(y.data_ptr()), self._stream())
        self._check(rc, "qv_upfirdn")
        return y

    def resample_rows(self, x, lengths, up: int, down: int, src_rows=None, out_pitch: int | None = None):
        """scipy.signal.resample_poly(row, up, down) for every row of a float32 cuda matrix in ONE launch (qv_upfirdn_batch):
        x [R0, pitch] holds the clips zero-padded, ``lengths`` their sample counts; ``src_rows`` (optional) picks which rows
        of x are resampled (output row r <- x[src_rows[r]], lengths[r] = that clip's length).  Returns (y [R, out_pitch]
        zero-padded like x, list of output lengths).  Every output sample is bit-identical to resample_poly's float32 one."""
        from .audio import resample_plan

        torch = self.torch
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        lens = np.ascontiguousarray(np.asarray(lengths, dtype=np.int64))
        rows = len(lens)
        plan = [resample_plan(up, down, int(n)) for n in sorted(set(lens.tolist()))]
        up_r, down_r, m0 = plan[0][0], plan[0][1], plan[0][3]
        if up_r == down_r == 1:
            y = x if src_rows is None else x[torch.as_tensor(list(src_rows), device=x.device)]
            return y.clone(), lens.tolist()
        # one filter for the whole batch: the longest zero padding any row's length asks for (padding taps add exact zeros)
        taps = max((p[2] for p in plan), key=len)
        n_out = [int((int(n) * up_r + down_r - 1) // down_r) for n in lens]
        pitch = int(out_pitch or max(n_out))
        assert pitch >= max(n_out)
        y = torch.empty((rows, pitch), dtype=torch.float32, device=x.device)
        src = None if src_rows is None else np.ascontiguousarray(np.asarray(list(src_rows), dtype=np.int32))
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
        rc = self.lib.qv_upfirdn_batch(self.h, C.c_void_p(x.data_ptr()), int(x.stride(0)), p(src), p(lens), rows, p(taps), len(taps),
                                       up_r, down_r, m0, C.c_void_p(y.data_ptr()), pitch, self._stream())
        self._check(rc, "qv_upfirdn_batch")
        return y, n_out

    def mixdown_rows(self, x, n_frames, channels: int):
        """interleaved [R, frames * channels] float32 cuda rows -> mono [R, max frames] (numpy's float32 mean over channels)"""
        torch = self.torch
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        nf = np.ascontiguousarray(np.asarray(n_frames, dtype=np.int64))
        y = torch.zeros((len(nf), int(nf.max())), dtype=torch.float32, device=x.device)
        rc = self.lib.qv_mixdown_batch(self.h, C.c_void_p(x.data_ptr()), int(x.stride(0)), nf.ctypes.data_as(C.c_void_p), len(nf), int(channels),
                                       C.c_void_p(y.data_ptr()), int(y.stride(0)), self._stream())
        self._check(rc, "qv_mixdown_batch")
        return y

    def speed_perturb_rows(self, x, lengths, factor: float, src_rows=None):
        """the TTA wrapper's speed perturbation (tta/run.py:60-71: up = int(factor * 10), down = 10) of many clips at once"""
        return self.resample_rows(x, lengths, int(factor * 10), 10, src_rows=src_rows)

    def speed_perturb(self, x, factor: float):
        """0.9 = 10 % slower, 1.1 = 10 % faster (tta/run.py:60-71: up = int(factor * 10), down = 10)."""
        return x if factor == 1.0 else self.resample_poly(x, int(factor * 10), 10)

    # ---------------------------------------------------------------- measurement
    GEMM_EPILOGUES = ["f16", "f16_swish", "f16_relu", "glu", "resid", "f32", "qkv"]

    def gemm_tiles(self, mode: int):
        """process-wide GEMM tile policy (qv_debug_gemm_tiles): 0 = 128-wide only, 1 = default, 2 = 256 x 256
        wherever the shape allows, -1 = environment / default."""
        self._check(self.lib.qv_debug_gemm_tiles(int(mode)), "qv_debug_gemm_tiles")

    def gemm_tile_height(self, mode: int):
        """tile height of the wide GEMM kernel (qv_debug_gemm_tile_height): 0 = 256 rows always, 1 = default (192 rows where
        they save a round of tiles and < 3 batches are in flight), 2 = that rule always, 3 = 192 rows always, -1 = env / default."""
        self._check(self.lib.qv_debug_gemm_tile_height(int(mode)), "qv_debug_gemm_tile_height")

    def weights_info(self) -> str:
        """precision mode and where the quantisation grids came from (qv_weights_info)"""
        buf = C.create_string_buffer(512)
        self._check(self.lib.qv_weights_info(self.h, buf, 512), "qv_weights_info")
        return buf.value.decode()

    def attention_variant(self, mode: int):
        """process-wide attention kernel variant (qv_debug_attention_variant): 3 = default (utterances of <= 128 frames on
        the single-pass kernel, longer ones on the loader-wave key-tiled kernel), 0 = that kernel with two heads per block for
        every utterance, 1 = one head per block, 2 = one wave per query tile, 4 = k_attention_x for every utterance,
        5 = single-pass + k_attention_x (0, 1, 2, 4: identical bits; 3, 5: identical bits), -1 = environment / default."""
        self._check(self.lib.qv_debug_attention_variant(int(mode)), "qv_debug_attention_variant")

    def kernel_variant(self, which: int, mode: int):
        """process-wide variant of one kernel (qv_debug_kernel_variant): which 0 = log-mel FFT (0 LDS, 1 registers),
        1 = precision 2's conv.0 (0 VALU, 1 f32 matrix pipe), 2 = span pass (0 one walk per span, 1 prefix-shared),
        3 = forward of a multi-context engine (0 plain launches, 1 hipGraph replay of a repeating shape);
        -1 = environment / default.  Identical bits either way."""
        self._check(self.lib.qv_debug_kernel_variant(int(which), int(mode)), "qv_debug_kernel_variant")

    def forward_graph_stats(self) -> dict:
        """forwards replayed as one hipGraph launch / graphs captured since the engine was created."""
        r, c = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.qv_debug_forward_graph_stats(self.h, C.byref(r), C.byref(c)), "qv_debug_forward_graph_stats")
        return {"replays": int(r.value), "captures": int(c.value)}

    def profile_gemm(self, enable: bool):
        self._check(self.lib.qv_profile_gemm(self.h, int(enable)), "qv_profile_gemm")

    def profile_gemm_read(self) -> list[dict]:
        """per GEMM kernel class: summed HIP-event ms, algorithmic FLOPs, launches."""
        ms = np.zeros(21)
        fl = np.zeros(21)
        n = np.zeros(21, np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        self._check(self.lib.qv_profile_gemm_read(self.h, p(ms), p(fl), p(n)), "qv_profile_gemm_read")
        out = []
        for c in range(21):
            if n[c]:
                epi, tile = self.GEMM_EPILOGUES[c // 3], c % 3
                out.append({"kernel": f"k_gemm256<{epi}>" if tile == 2 else f"k_gemm<{epi},{128 if tile else 64}>",
                            "ms": float(ms[c]), "flops": float(fl[c]), "launches": int(n[c])})
        return out

    def inject_logprobs(self, log_probs=None, t_frames=None):
        """measurement hook (qv_profile_inject_logprobs): the post-logits stages of predict_batch_async read this cuda
        tensor [B, t_max, 1025] instead of the forward's output (the forward still runs); None clears it.  The tensor is
        kept alive by the engine object."""
        if log_probs is None:
            self._injected = None
            self._check(self.lib.qv_profile_inject_logprobs(self.h, None, 0, None, 0), "qv_profile_inject_logprobs")
            return
        assert log_probs.is_cuda and log_probs.dtype == self.torch.float32 and log_probs.is_contiguous() and log_probs.shape[2] == 1025
        t = np.ascontiguousarray(np.asarray(t_frames, dtype=np.int32))
        self._injected = log_probs
        self._check(self.lib.qv_profile_inject_logprobs(self.h, C.c_void_p(log_probs.data_ptr()), log_probs.shape[1],
                                                        t.ctypes.data_as(C.c_void_p), log_probs.shape[0]), "qv_profile_inject_logprobs")

    def profile_stages(self, enable: bool):
        """device-side stage timers (the reference's C2C_DIRECT_MIXED_PROFILE split, mixed/run.py:117-124)"""
        self._check(self.lib.qv_profile_stages(self.h, int(enable)), "qv_profile_stages")

    def stage_times(self, ctx: int | None = None) -> dict:
        """seconds spent in forward / decode / build / rerank by the last batch of context `ctx`
        (default: the most recent call's); waits for that batch."""
        if ctx is None:
            ctx = int(self.lib.qv_last_context(self.h))
        ms = np.zeros(4, np.float32)
        self._check(self.lib.qv_stage_times(self.h, ctx, ms.ctypes.data_as(C.c_void_p)), "qv_stage_times")
        return {k: float(v) * 1e-3 for k, v in zip(("forward", "decode", "build", "rerank"), ms)}

    REPLAY_SHAPES = ["FFN-up [M,512]x[512,2048]+Swish", "FFN-down [M,2048]x[2048,512]+residual", "QKV [M,512]x[512,1536]",
                     "attention out [M,512]x[512,512]+residual", "pointwise conv [M,512]x[512,1024]+GLU"]

    def replay_gemm(self, which: int, iters: int = 50) -> dict:
        """average duration (HIP events on the launch stream) of `iters` back-to-back launches of one
        layer-0 GEMM with the shapes of the last forward."""
        us, fl = C.c_double(), C.c_double()
        self._check(self.lib.qv_profile_replay_gemm(self.h, which, iters, C.byref(us), C.byref(fl), self._stream()),
                    "qv_profile_replay_gemm")
        name = C.create_string_buffer(64)
        self._check(self.lib.qv_profile_replay_kernel(self.h, which, name, 64), "qv_profile_replay_kernel")
        return {"kernel": name.value.decode(), "shape": self.REPLAY_SHAPES[which], "avg_us": us.value,
                "flops": fl.value, "launches": iters}

    # ---------------------------------------------------------------- debug ------
    # ---------------------------------------------------------------- streaming row
    def next_verse(self, surah: int, ayah: int) -> int:
        """Global index of the verse after (surah, ayah) in mushaf order, -1 if there is none
        or (surah, ayah) does not exist (QuranDB.get_next_verse, shared/quran_db.py:81-90)."""
        t = self.tables.s
        if not (1 <= surah <= 114) or not (1 <= ayah <= int(t["surah_len"][surah - 1])):
            return -1
        idx = int(t["surah_start"][surah - 1]) + ayah - 1
        return idx + 1 if idx + 1 < self.tables.n_verses else -1

    def track_match(self, texts, last_refs=None) -> list[dict | None]:
        """VerseTracker._find_best_match for a batch of accumulated texts in one launch
        (qv_tracker_match).  last_refs[b] = (surah, ayah) of the tracker's last emission or None.
        Returns per text None (no verse scores above 0) or
        {surah, ayah, n_words, score, verse, variant}; the caller applies its emit gates."""
        n = len(texts)
        if n == 0:
            return []
        last_refs = last_refs or [None] * n
        texts = [front_window(t) for t in texts]   # > 1,024 characters: matched on the front window
        enc = [self.tables.encode(t) for t in texts]
        off = np.zeros(n + 1, np.int32)
        off[1:] = np.cumsum([len(e) for e in enc])
        codes = np.ascontiguousarray(np.concatenate(enc) if off[-1] else np.zeros(1, np.uint8))
        nw = np.ascontiguousarray(np.array([len(t.split()) for t in texts], np.int32))
        bonus = np.ascontiguousarray(np.array([self.next_verse(*r) if r else -1 for r in last_refs], np.int32))
        out = (QvTrackMatch * n)()
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        rc = self.lib.qv_tracker_match(self.h, p(codes), p(off), p(nw), p(bonus), n, C.cast(out, C.c_void_p),
                                       self._stream())
        self._check(rc, "qv_tracker_match")
        return [None if m.verse < 0 else
                {"surah": m.surah, "ayah": m.ayah, "n_words": m.n_words, "score": m.score, "verse": m.verse,
                 "variant": m.variant} for m in out]

    def continuation_bonuses(self, hint) -> list[tuple[int, float]]:
        """QuranDB._continuation_bonuses (shared/quran_db.py:121-146) as (verse index, bonus)
        pairs: the three ayat after the hint, or - when the hint is the last ayah of its surah -
        the first three of the next surah."""
        if not hint:
            return []
        s, a = hint
        t = self.tables.s
        n_surah = len(t["surah_len"])

        def idx(su, ay):
            if 1 <= su <= n_surah and 1 <= ay <= int(t["surah_len"][su - 1]):
                return int(t["surah_start"][su - 1]) + ay - 1
            return None

        out = []
        if idx(s, a + 1) is not None:
            for k, bonus in enumerate((0.22, 0.12, 0.06)):
                i = idx(s, a + 1 + k)
                if i is not None:
                    out.append((i, bonus))
        elif 1 <= s + 1 <= n_surah:
            first = int(t["surah_start"][s])
            for k, bonus in zip(range(min(3, int(t["surah_len"][s]))), (0.22, 0.12, 0.06)):
                out.append((first + k, bonus))
        return out

    def match_verse(self, text: str, threshold: float = 0.3, max_span: int = 3, hint=None) -> dict | None:
        """QuranDB.match_verse without the trigram restriction (qv_match_verse).  Returns None
        below the threshold, else {surah, ayah, ayah_end (None for one ayah), score, n_words}
        where n_words counts the words of the matched text_clean (what the caller trims by)."""
        text = normalize_arabic(text)
        if not text.strip():
            return None
        codes = np.ascontiguousarray(self.tables.encode(front_window(text)))
        bon = self.continuation_bonuses(hint)
        bv = np.ascontiguousarray(np.array([b[0] for b in bon] + [0] * (3 - len(bon)), np.int32))
        bb = np.ascontiguousarray(np.array([b[1] for b in bon] + [0.0] * (3 - len(bon)), np.float64))
        st, sp, sc = C.c_int32(), C.c_int32(), C.c_double()
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        rc = self.lib.qv_match_verse(self.h, p(codes), len(codes), len(bon), p(bv), p(bb), int(max_span),
                                     C.byref(st), C.byref(sp), C.byref(sc), self._stream())
        self._check(rc, "qv_match_verse")
        if st.value < 0 or not (sc.value >= threshold):
            return None
        v, span = st.value, sp.value
        t = self.tables.s
        if span == 1:
            n_words = int(t["clean_nw"][v])
        else:
            first = int(t["nobsm_nw"][v]) or int(t["clean_nw"][v])
            n_words = first + sum(int(t["clean_nw"][v + k]) for k in range(1, span))
        s, a = int(self.tables.surah[v]), int(self.tables.ayah[v])
        return {"surah": s, "ayah": a, "ayah_end": a + span - 1 if span > 1 else None, "score": sc.value,
                "n_words": n_words, "verse": v, "span": span}

    def transcribe_batch(self, audio, lengths) -> list[str]:
        """Forward + greedy CTC decode of a zero-padded batch (float32 cuda [B, N]); the
        transcript of each row as c2c-direct/run.py:187-204 returns it."""
        lp, T = self.forward(audio, lengths)
        ids = lp.argmax(-1).cpu().numpy()
        out = []
        for b, t in enumerate(T):
            row = ids[b, :t]
            keep = np.ones(t, bool)
            keep[1:] = row[1:] != row[:-1]
            out.append(self.transcript_of(row[keep & (row != 1024)].tolist()))
        return out

    def debug_retrieve(self, transcript: str) -> dict:
        """match_verse + candidate assembly for an already-normalised transcript."""
        codes = np.ascontiguousarray(self.tables.encode(transcript))
        cap = 2048
        bs, bp, nc, nr = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        bsc = C.c_double()
        cs = np.zeros(cap, np.int32)
        cp = np.zeros(cap, np.int32)
        csc = np.zeros(cap, np.float64)
        ri = np.zeros(128, np.int32)
        rs = np.zeros(128, np.float64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        rc = self.lib.qv_debug_retrieve(self.h, p(codes), len(codes), C.byref(bs), C.byref(bp), C.byref(bsc),
                                        p(cs), p(cp), p(csc), cap, C.byref(nc), p(ri), p(rs), C.byref(nr),
                                        self._stream())
        self._check(rc, "qv_debug_retrieve")
        n = min(nc.value, cap)
        return {"base_start": bs.value, "base_span": bp.value, "base_score": bsc.value,
                "cand_start": cs[:n].copy(), "cand_span": cp[:n].copy(), "cand_score": csc[:n].copy(),
                "runner_idx": ri[: nr.value].copy(), "runner_score": rs[: nr.value].copy()}

    def debug_ctc_loss(self, log_probs_2d, id_lists) -> np.ndarray:
        assert log_probs_2d.is_cuda and log_probs_2d.dim() == 2 and log_probs_2d.is_contiguous()
        tg = np.ascontiguousarray(np.concatenate([np.asarray(x, np.uint16) for x in id_lists]))
        lens = np.ascontiguousarray(np.array([len(x) for x in id_lists], np.int32))
        out = np.zeros(len(id_lists), np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        rc = self.lib.qv_debug_ctc_loss(self.h, C.c_void_p(log_probs_2d.data_ptr()), log_probs_2d.shape[0],
                                        p(tg), p(lens), len(id_lists), p(out), self._stream())
        self._check(rc, "qv_debug_ctc_loss")
        return out

    def forward_tap(self, what: int, layer: int, shape):
        out = self.torch.empty(shape, dtype=self.torch.float32, device=f"cuda:{self.device}")
        rc = self.lib.qv_debug_forward_tap(self.h, what, layer, C.c_void_p(out.data_ptr()), self._stream())
        self._check(rc, "qv_debug_forward_tap")
        return out


_hip = None


def _hip_runtime():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    return _hip
