"""Deterministic synthetic inputs shared by the golden generator, the tests and bench.py.

Everything here is integer hashing + exact IEEE multiplies/adds (no libm calls), so
the same recipe yields bit-identical arrays in the build container and on the GPU box.
Idiom follows the reference's own TS test that synthesises ``[T,V]`` log-probs from
a token path (web/frontend/test/trie-beam.test.ts:105-136): token / blank frames with
the path entry boosted above noise.
"""

from __future__ import annotations

import numpy as np

VOCAB = 1025
BLANK = 1024


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_noise(shape, seed: int) -> np.ndarray:
    """float32 array, zero-mean, unit-ish variance (Irwin-Hall of 4 bytes), exact."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
        h = _splitmix64(idx)
    s = (
        (h & np.uint64(0xFF))
        + ((h >> np.uint64(8)) & np.uint64(0xFF))
        + ((h >> np.uint64(16)) & np.uint64(0xFF))
        + ((h >> np.uint64(24)) & np.uint64(0xFF))
    ).astype(np.int64) - 510
    return (s.astype(np.float32) * np.float32(1.0 / 147.8)).reshape(shape)


def frame_path(ids, T: int, rep: int = 2):
    """token path -> per-frame ids (rep frames per token, one blank between, blank pad)."""
    path = []
    for tok in ids:
        path.extend([int(tok)] * rep)
        path.append(BLANK)
    path = path[:T]
    path += [BLANK] * (T - len(path))
    return np.asarray(path, dtype=np.int64)


def synth_logits(ids, T: int, seed: int, noise: float, boost: float, rep: int = 2) -> np.ndarray:
    """float32 [T,1025] logits; caller applies log_softmax (torch, float32)."""
    lg = hash_noise((T, VOCAB), seed) * np.float32(noise)
    if boost != 0.0:
        p = frame_path(ids, T, rep)
        lg[np.arange(T), p] += np.float32(boost)
    elif noise == 0.0:
        lg[:, BLANK] += np.float32(5.0)
    return np.ascontiguousarray(lg, dtype=np.float32)


def synth_audio(B: int, N: int, seed: int = 20260630) -> np.ndarray:
    """SURVEY.md section 8(d) synthetic clip: noise + 3 tones x 4 Hz envelope, clipped."""
    t = np.arange(N, dtype=np.float64) / 16000.0
    env = 0.5 - 0.5 * np.cos(2 * np.pi * 4.0 * t)
    tones = 0.1 * (np.sin(2 * np.pi * 220 * t) + np.sin(2 * np.pi * 440 * t) + np.sin(2 * np.pi * 880 * t))
    base = (tones * env).astype(np.float32)
    out = np.empty((B, N), dtype=np.float32)
    for b in range(B):
        out[b] = base * np.float32(1.0 + 0.05 * (b % 7)) + np.float32(0.05) * hash_noise((N,), seed + b)
    return np.clip(out, -1.0, 1.0)
