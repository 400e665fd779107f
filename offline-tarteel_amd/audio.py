"""Audio ingest on the host side of the boundary (reference: shared/audio.py:8-18 ``load_audio``
and experiments/c2c-direct-mixed-tta/run.py:60-71 ``_speed_perturb``).

``load_audio(path, sr=16000)`` -> float32 mono at 16 kHz.  The reference leans on
librosa/soundfile; this image has neither (and no ffmpeg), so WAV containers are parsed with the
standard library (PCM 8/16/24/32-bit and IEEE float) and resampled with
``scipy.signal.resample_poly``.  librosa's default resampler is soxr_hq, a different FIR design:
for non-16 kHz sources the samples agree to resampler-design tolerance, not bit-for-bit
(documented in DESIGN.md); 16 kHz sources (the 29 RetaSy files of the v1 corpus) are exact.
Compressed formats (mp3/m4a) raise: there is no decoder in this environment.

``load_audio_device(paths, eng)`` (round 6) is the same computation with the mix-down and the polyphase FIR on the GPU
(qv_mixdown_batch / qv_upfirdn_batch); the plugin's batched entry uses it (QVERSE_INGEST=host switches back).
"""

from __future__ import annotations

import struct
from fractions import Fraction
from pathlib import Path

import numpy as np

TARGET_SR = 16000


def _decode_wav(path: Path):
    """-> (interleaved float32 samples, channels, sample rate): the container's PCM as it is, no mix-down, no resampling"""
    data = path.read_bytes()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file (no decoder for compressed audio in this environment)")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8: pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:  # WAVE_FORMAT_EXTENSIBLE -> sub-format
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: malformed WAV")
    tag, ch, sr, bits = fmt
    if tag == 3 and bits == 32:
        x = np.frombuffer(pcm, dtype="<f4").astype(np.float32)
    elif tag == 3 and bits == 64:
        x = np.frombuffer(pcm, dtype="<f8").astype(np.float32)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 8:
        x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(pcm, dtype="<i4").astype(np.float32) / 2147483648.0
    elif tag == 1 and bits == 24:
        b = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        x = v.astype(np.float32) / 8388608.0
    else:
        raise ValueError(f"{path}: unsupported WAV encoding tag={tag} bits={bits}")
    ch = max(1, int(ch))
    return np.ascontiguousarray(x[: len(x) // ch * ch], dtype=np.float32), ch, sr


def _read_wav(path: Path):
    x, ch, sr = _decode_wav(path)
    if ch > 1:
        x = x.reshape(-1, ch).mean(axis=1)
    return x.astype(np.float32), sr


def probe_samples(path: str, sr: int = TARGET_SR) -> int:
    """Length of a file in samples at ``sr`` WITHOUT decoding it (RIFF header walk): what the sharded runner sorts by
    (dist.shard_plan).  Unknown containers fall back to the file size -- any monotone proxy of the duration will do."""
    p = Path(path)
    try:
        with open(p, "rb") as f:
            head = f.read(12)
            if len(head) == 12 and head[:4] == b"RIFF" and head[8:12] == b"WAVE":
                fmt = None
                while True:
                    ch = f.read(8)
                    if len(ch) < 8:
                        break
                    cid, size = ch[:4], struct.unpack("<I", ch[4:8])[0]
                    if cid == b"fmt ":
                        body = f.read(size + (size & 1))
                        _, nch, native, _, align, _ = struct.unpack("<HHIIHH", body[:16])
                        fmt = (native, max(1, align))
                    elif cid == b"data":
                        if fmt:
                            return int(size // fmt[1] * sr // max(1, fmt[0]))
                        break
                    else:
                        f.seek(size + (size & 1), 1)
        return int(p.stat().st_size)
    except OSError:
        return 0


def resample(audio: np.ndarray, orig_sr: int, target_sr: int) -> np.ndarray:
    if orig_sr == target_sr:
        return audio.astype(np.float32)
    from scipy.signal import resample_poly

    fr = Fraction(target_sr, orig_sr)
    return resample_poly(audio, fr.numerator, fr.denominator).astype(np.float32)


def load_audio(path: str, sr: int = TARGET_SR) -> np.ndarray:
    audio, native = _read_wav(Path(path))
    return resample(audio, native, sr)


def load_audio_device(paths, eng, sr: int = TARGET_SR):
    """load_audio for a list of files with the mix-down and the resampling ON THE GPU (reference: shared/audio.py:8-18).

    The host only parses the containers (PCM -> float32, interleaved).  Files that are already mono at ``sr`` are
    uploaded as they are; every other (sample rate, channel count) group is uploaded interleaved, mixed down by
    qv_mixdown_batch (numpy's float32 mean over the channels) and resampled by qv_upfirdn_batch with the rational
    factor sr / native (160/441 for 44.1 kHz, 1/3 for 48 kHz, ...): the float32 polyphase FIR of
    scipy.signal.resample_poly, bit for bit -- i.e. exactly what load_audio() computes on the host, which differs
    from the reference's librosa default (soxr_hq, another FIR design) by resampler-design tolerance only
    (tools/resample_delta.py).  Returns (float32 cuda tensor [len(paths), max length] zero-padded, list of lengths)."""
    import torch

    dev = torch.device(f"cuda:{eng.device}")
    decoded = [_decode_wav(Path(p)) for p in paths]
    groups: dict = {}
    for i, (x, ch, native) in enumerate(decoded):
        groups.setdefault((native, ch), []).append(i)
    rows_of: dict = {}
    for (native, ch), idx in groups.items():
        frames = [len(decoded[i][0]) // ch for i in idx]
        if min(frames) < 1:
            raise ValueError(f"{paths[idx[frames.index(min(frames))]]}: empty audio")
        host = np.zeros((len(idx), max(frames) * ch), dtype=np.float32)
        for r, i in enumerate(idx):
            host[r, : frames[r] * ch] = decoded[i][0]
        x = torch.from_numpy(host).to(dev, non_blocking=False)
        if ch > 1:
            x = eng.mixdown_rows(x, frames, ch)
        lens = frames
        if native != sr:
            fr = Fraction(sr, native)
            x, lens = eng.resample_rows(x, frames, fr.numerator, fr.denominator)
        for r, i in enumerate(idx):
            rows_of[i] = (x, r, int(lens[r]))
    lens_out = [rows_of[i][2] for i in range(len(paths))]
    out = torch.zeros((len(paths), max(lens_out)), dtype=torch.float32, device=dev)
    for i in range(len(paths)):
        x, r, n = rows_of[i]
        out[i, :n] = x[r, :n]
    return out, lens_out


_PLAN_CACHE: dict = {}


def resample_plan(up: int, down: int, n_in: int):
    """Host half of resample_poly for float32 input: (up, down, taps float32, first kept output
    index, n_out).  The taps are what scipy hands to upfirdn -- firwin(2*half_len+1, 1/max_rate,
    window=("kaiser", 5.0)) cast to float32, scaled by `up` in float32, zero-padded in front so
    that output sample 0 is centred and behind until the FIR produces enough samples.  The FIR
    itself runs on the GPU (Engine.resample_poly -> qv_upfirdn)."""
    import math

    g = math.gcd(int(up), int(down))
    up, down = int(up) // g, int(down) // g
    if up == down == 1:
        return 1, 1, None, 0, n_in      # resample_poly returns a copy
    n_out = (n_in * up + down - 1) // down
    max_rate = max(up, down)
    half_len = 10 * max_rate
    key = (up, down)
    if key not in _PLAN_CACHE:
        from scipy.signal import firwin

        h = firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0)).astype(np.float32)
        h *= up
        _PLAN_CACHE[key] = h
    h = _PLAN_CACHE[key]
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down

    def out_len(len_h: int) -> int:
        nt = (n_in + (len_h + (-len_h % up)) // up - 1) * up
        return nt // down + (1 if nt % down else 0)

    n_post_pad = 0
    while out_len(len(h) + n_pre_pad + n_post_pad) < n_out + n_pre_remove:
        n_post_pad += 1
    taps = np.concatenate((np.zeros(n_pre_pad, np.float32), h, np.zeros(n_post_pad, np.float32)))
    return up, down, np.ascontiguousarray(taps), n_pre_remove, n_out


def speed_perturb(audio_16k: np.ndarray, factor: float) -> np.ndarray:
    """0.9 = 10 % slower, 1.1 = 10 % faster: resample_poly(x, int(factor*10), 10), default Kaiser
    FIR -- the same scipy call the reference's TTA wrapper makes."""
    if factor == 1.0:
        return audio_16k
    from scipy.signal import resample_poly

    return resample_poly(audio_16k, int(factor * 10), 10).astype("float32")
