"""The reproducibility floor of the reference's quantised arithmetic, measured per clip.

onnxruntime quantises the activations of every Conv per call (DynamicQuantizeLinear): ~60 rounding discontinuities sit
between the audio and the log-probs, and a perturbation far below any tolerance flips a few values across a rounding
boundary.  `oracle_floor` runs the CPU oracle (oracle/fastconformer_ref.py::OrtMixed) against ITSELF under changes
that are not errors -- float32 summation order (1 thread instead of N) and < 1 ulp of multiplicative noise on every
Linear input (two seeds) -- on the SAME clip the device result is judged on, and returns the envelope.  The device
must then be no further from the oracle than `k` x that envelope (tests/test_gpu_ort_mixed.py, __graft_entry__.smoke):
a bound that tightens by itself wherever the arithmetic is more reproducible (structured weights, a trained model)."""

from __future__ import annotations

import torch


def delta(lp, lp_ref, T):
    """(max |d|, rms d, argmax agreement) over the valid frames of a batch"""
    d = torch.cat([(lp[b, : T[b]].float().cpu() - lp_ref[b, : T[b]]).flatten() for b in range(len(T))])
    same = sum(int((lp[b, : T[b]].cpu().argmax(-1) == lp_ref[b, : T[b]].argmax(-1)).sum()) for b in range(len(T))) / max(1, sum(T))
    return float(d.abs().max()), float(d.pow(2).mean().sqrt()), same


def make_noise_ops(R, eps: float, seed: int, **ort_kw):
    class Noise(R.OrtMixed):
        def __init__(self):
            super().__init__(**ort_kw)
            self.g = torch.Generator().manual_seed(seed)

        def linear(self, w, name, x, bias_name):
            x = x * (1 + eps * (torch.rand(x.shape, generator=self.g) * 2 - 1))
            return super().linear(w, name, x, bias_name)

    return Noise()


def make_f16_input_ops(R, **ort_kw):
    """the device's design point as a perturbation of the ORACLE: every Linear input rounded to float16 (what an f16-operand
    MFMA GEMM sees); quantisers and integer convolutions stay exact"""
    class F16(R.OrtMixed):
        def linear(self, w, name, x, bias_name):
            return super().linear(w, name, x.half().float(), bias_name)

    return F16(**ort_kw)


def oracle_floor(R, w, audio, lens, lp_ref, T, threads: int | None = None, seeds=(1, 2), one_thread: bool = True,
                 f16_inputs: bool = False, **ort_kw):
    """envelope of oracle-vs-oracle over: 1 intra-op thread (when `one_thread`), 1e-7 relative noise on the Linear inputs
    (one run per seed), and -- only when `f16_inputs` -- the Linear inputs rounded to float16.  lp_ref = the oracle's result with `threads` threads (the caller's reference).
    Returns {"max": .., "rms": .., "argmax": .., "rows": {...}}."""
    threads = threads or torch.get_num_threads()
    rows = {}
    if one_thread:
        torch.set_num_threads(1)
        try:
            rows["threads_1"] = delta(R.forward(w, audio, lens, ort=R.OrtMixed(**ort_kw))[0], lp_ref, T)
        finally:
            torch.set_num_threads(threads)
    for s in seeds:
        rows[f"ulp_noise_seed{s}"] = delta(R.forward(w, audio, lens, ort=make_noise_ops(R, 1e-7, s, **ort_kw))[0], lp_ref, T)
    if f16_inputs:
        rows["f16_linear_inputs"] = delta(R.forward(w, audio, lens, ort=make_f16_input_ops(R, **ort_kw))[0], lp_ref, T)
    return {"max": max(r[0] for r in rows.values()), "rms": max(r[1] for r in rows.values()),
            "argmax": min(r[2] for r in rows.values()), "rows": rows}
