#!/usr/bin/env python
"""How much does the ingest resampler matter?  (shared/audio.py:8-18 resamples with librosa,
whose default is soxr_hq; this repo's audio.py uses scipy.signal.resample_poly's Kaiser(5.0) FIR -- soxr is absent.)

    python tools/resample_delta.py [--seconds 6] [--out profiles/rNN_resample_delta.json]

A 44.1 kHz synthetic speech-like clip (harmonic stack with vibrato under a syllable envelope, shaped noise bursts,
energy up to 12 kHz, so there IS content above the new Nyquist to alias) is brought to 16 kHz by
  ref     a 16,001-tap Kaiser(beta 14.8, > 140 dB) windowed-sinc polyphase resampler in float64 -- a stand-in for
          "a high-quality resampler" (soxr_hq's stop band is ~ -125 dB; both are transparent at float32 precision)
  poly    audio.resample (scipy.signal.resample_poly defaults), what load_audio does
and the two are compared as signals (SNR), as normalised log-mel features, and through the fp32 oracle forward on
seeded weights (max |delta log-prob|, argmax agreement).  With random weights "transcript equality" (the reference's
own bar for a re-implemented front end) can only be read as argmax agreement.
"""
from __future__ import annotations

import argparse
import json
import sys
from fractions import Fraction
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def speechlike(n: int, sr: int, seed: int = 5):
    import numpy as np

    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    f0 = 120 * (1 + 0.08 * np.sin(2 * np.pi * 5.3 * t)) * (1 + 0.3 * np.sin(2 * np.pi * 0.7 * t))
    phase = 2 * np.pi * np.cumsum(f0) / sr
    env = np.clip(np.sin(2 * np.pi * 3.1 * t), 0, None) ** 2
    x = sum((1.0 / (1 + 0.12 * k)) * np.sin(k * phase + rng.uniform(0, 6.28)) for k in range(1, 90))   # harmonics to ~12 kHz
    noise = rng.normal(size=n) * (np.clip(np.sin(2 * np.pi * 3.1 * t + 2.0), 0, None) ** 4)
    x = 0.05 * x * env + 0.03 * noise
    return (x / np.abs(x).max() * 0.7).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import numpy as np
    import torch

    import offline_tarteel_amd  # noqa: F401
    from offline_tarteel_amd.audio import resample
    from oracle import fastconformer_ref as R

    sr = 44100
    x = speechlike(int(args.seconds * sr), sr)
    fr = Fraction(16000, sr)
    up, down = fr.numerator, fr.denominator          # 160 / 441
    poly = resample(x, sr, 16000)
    half = 441 * 18
    from scipy.signal import firwin, upfirdn
    h = firwin(2 * half + 1, 1.0 / down, window=("kaiser", 14.8)) * up
    ref = upfirdn(h, x.astype(np.float64), up, down)[half // down: half // down + len(poly)].astype(np.float32)
    n = min(len(ref), len(poly))
    ref, poly = ref[:n], poly[:n]
    err = poly.astype(np.float64) - ref
    snr = 10 * np.log10(float((ref.astype(np.float64) ** 2).sum() / (err ** 2).sum()))
    torch.set_num_threads(8)
    a = torch.from_numpy(np.stack([ref, poly]))
    feats, tm = R.frontend(a, torch.tensor([n, n]))
    dmel = float((feats[0] - feats[1]).abs().max())
    w = R.random_weights(20260630)
    lp, T = R.forward(w, a, [n, n])
    t = int(T[0])
    dlp = float((lp[0, :t] - lp[1, :t]).abs().max())
    agree = float((lp[0, :t].argmax(-1) == lp[1, :t].argmax(-1)).float().mean())
    doc = {"what": "scipy.signal.resample_poly (Kaiser 5.0 default, audio.resample) vs a > 140 dB windowed-sinc resampler, "
                   "44.1 kHz -> 16 kHz, synthetic speech-like clip of %g s" % args.seconds,
           "snr_db": round(snr, 1), "max_abs_sample_error": float(np.abs(err).max()),
           "max_abs_normalised_logmel_delta": round(dmel, 4), "max_abs_delta_logprob_fp32_oracle_seeded_weights": round(dlp, 4),
           "argmax_agreement": round(agree, 4), "frames": t}
    print(json.dumps(doc, indent=1))
    if args.out:
        Path(args.out).write_text(json.dumps(doc, indent=1) + "\n")


if __name__ == "__main__":
    main()
