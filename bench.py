#!/usr/bin/env python
"""bench.py -- utterances/sec of the c2c-direct-mixed hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (FastConformer-CTC forward -> greedy decode ->
verse retrieval -> CTC rerank) over one batch of synthetic 16 kHz clips that is already resident
in HBM.  Workloads (BASELINE.json configs):
  default, N < 8   64 clips x 10 s per GPU, fp16 weights          configs[1] (the headline line)
  --precision mixed --batch 256   int4 Linear + int8 pointwise-conv weights   configs[2]
  default, N = 8   256 clips x 10 s per GPU = 2048 per step       configs[3]
  --workload tta30 64 clips x 30 s per GPU through the TTA wrapper (anchor pass, 0.5 gate, GPU
                   0.9x / 1.1x copies of the gated clips, majority / best-score pick)   configs[4]
For N > 1 every rank processes its own batch (utterances are independent: pure data parallel,
weak scaling) and the packed (surah, ayah, ayah_end, score) rows of every batch are all-gathered
over RCCL.

The engine keeps --contexts (default 4) batches in flight on internal streams (and asks the HIP runtime for
8 hardware queues, GPU_MAX_HW_QUEUES, so that every stream gets its own: with the runtime's default of 4, four
context streams + the caller's share queues and lose 12 %; profiles/r02_h_contexts_hwq_sweep.txt): each step still
runs the whole path on its own batch of 64, but the latency-bound post-logits kernels of one
batch execute under the forward pass of the next (DESIGN.md "Batches in flight"; --contexts 1
gives the one-batch-at-a-time figure).  The timed region ends after every batch has finished
(device synchronise) and, for N > 1, after every batch's rows have been gathered.

Prints ONE JSON line on rank 0 (contract in the task description) including
  roofline      dominant kernel (the FFN-up GEMM class) measured with HIP events on its stream
  cpu_baseline  the CPU oracle (fp32 PyTorch forward + C restatement of the post-logits
                stages) timed on this node's host cores on a bounded sample of the same clips
  post_logits   the post-logits stages alone on verse-shaped log-probs (the random-weight clips
                decode to near-empty transcripts and never exercise retrieval): ms per batch
                with every transcript passing the 0.80 text gate and with every one failing it
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

# one hardware queue per stream (4 context streams + the caller's + copies); read by the HIP runtime when it initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # kernel arguments in device memory (launch latency)
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

PEAK_F16_TFLOPS = 2500.0     # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md ("~2.5 PF dense")
FLOP_PER_UTT_10S = 28.5e9    # algorithmic forward FLOPs of one 10 s utterance


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=0,
                    help="utterances per GPU per step (default 64; 256 with --gpus 8 = BASELINE configs[3])")
    ap.add_argument("--seconds", type=float, default=0.0, help="clip length (default 10; 30 for --workload tta30)")
    ap.add_argument("--workload", choices=("clips", "tta30", "strong2048"), default="clips",
                    help="clips: the plain hot path; tta30: c2c-direct-mixed-tta on 30 s clips (configs[4]); strong2048: ONE ragged "
                         "global batch of 2,048 clips of 5-30 s dealt to the ranks by dist.shard_plan (strong scaling: a step is the "
                         "whole global batch, the rows come back through dist.all_gather_results)")
    ap.add_argument("--deal", choices=("strided", "contiguous"), default="strided", help="strong2048: how shard_plan deals the sorted clips")
    ap.add_argument("--tta-mix", action="store_true",
                    help="tta30 only: the anchor pass's post-logits stages read verse-shaped log-probs at the reference's branch "
                         "ratio (7 of its 53 v1 clips scored < 0.80; qv_profile_inject_logprobs), so that about one clip in eight "
                         "fails the 0.5 TTA gate instead of every clip; the forward still runs on every clip")
    ap.add_argument("--no-post-logits", action="store_true", help="skip the verse-shaped post-logits replay legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the short secondary legs (BASELINE configs[2] and configs[4] workloads) of the default run")
    ap.add_argument("--cpu-sample", type=int, default=24, help="clips timed on the host for cpu_baseline")
    ap.add_argument("--literal", action="store_true", help="run search()/pass-3 even when the gate passes")
    ap.add_argument("--capacity-seconds", type=float, default=0.0,
                    help="engine capacity (max_samples) in seconds, default = the clip length: a production engine is created for "
                         "the longest clip it may ever see (the plugin's default is 30 s) and must not pay for it on shorter ones")
    ap.add_argument("--contexts", type=int, default=4,
                    help="batches in flight per GPU (execution contexts of the engine, 1..8)")
    ap.add_argument("--precision", choices=("fp16", "mixed", "ort"), default="fp16",
                    help="fp16 = BASELINE configs[1] (the headline line); mixed = int4 Linear + int8 pointwise-conv weights "
                         "dequantised to f16 operands (W4A16 / W8A16); ort = the arithmetic onnxruntime runs on the reference's "
                         "file: int4 Linear + DynamicQuantizeLinear / ConvInteger (uint8 activations, int8 weights, i8 MFMA) "
                         "on every Conv (QV_PREC_ORT_MIXED)")
    a = ap.parse_args()
    if a.batch <= 0:
        a.batch = 256 if (a.gpus >= 8 and a.workload == "clips") else 64
    if a.workload == "strong2048":
        a.seconds = 30.0
    if a.seconds <= 0:
        a.seconds = 30.0 if a.workload == "tta30" else 10.0
    return a


def post_logits_legs(eng, B: int, T: int, steps: int = 10):
    """SURVEY.md 8(d) replay workload: log-probs synthesised from the token ids of seeded verses (the
    tests' recipe), once clean enough that every transcript passes the 0.80 text gate (0 % use_ctc) and
    once corrupted so that every one fails it (100 %: search over all verses, pass 3, candidate spans,
    CTC rerank).  Returns ms per batch of B for both (median, mean and worst of `steps` synchronous calls)."""
    import numpy as np
    import torch

    from synth import synth_logits

    rng = np.random.default_rng(20260630)
    n_verses = len(eng.tables.s["tok_off"]) // 6
    out = {"batch": B, "frames": T}
    for key, noise, boost in (("gate_pass", 1.0, 8.0), ("gate_fail", 3.5, 4.0)):
        lps, used = [], 0
        while len(lps) < B:
            v = int(rng.integers(0, n_verses))
            ids = eng.tables.token_ids(v, 1).tolist()
            if not (4 <= len(ids) and 2 * len(ids) + 1 <= T):
                continue
            lg = torch.from_numpy(synth_logits(ids, T, seed=1000 + used, noise=noise, boost=boost, rep=2))
            lps.append(torch.log_softmax(lg, -1))
            used += 1
        lp = torch.stack(lps).cuda(eng.device).contiguous()
        res = eng.decode_retrieve_rerank(lp, [T] * B, want_text=False)
        for _ in range(2):
            eng.decode_retrieve_rerank(lp, [T] * B, want_text=False)
        torch.cuda.synchronize()
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            eng.decode_retrieve_rerank(lp, [T] * B, want_text=False)  # synchronous: returns with the rows on the host
            ts.append((time.perf_counter() - t0) * 1e3)
        # median: the HIP runtime occasionally stalls one call for tens of ms (pool growth; seen once in ~300
        # calls on an idle box, independent of the engine's state); the mean and the worst call are kept beside it
        out[key] = {"ms_per_batch": round(sorted(ts)[len(ts) // 2], 3), "ms_per_batch_mean": round(sum(ts) / len(ts), 3),
                    "ms_per_batch_max": round(max(ts), 3),
                    "use_ctc_fraction": round(sum(r["use_ctc"] for r in res) / B, 3),
                    "mean_candidates": round(sum(r["n_candidates"] for r in res) / B, 1)}
    return out


def host_cores():
    """(logical CPUs, physical cores) of this node -- SURVEY.md 8(d) asks for both next to the CPU line."""
    logical = os.cpu_count() or 1
    phys = set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return logical, (len(phys) or logical)


def post_logits_cpu_split(orc, lps, budget_s: float = 8.0):
    """the C restatement of the post-logits stages on the host: one thread, and one thread per utterance on all cores
    (ctypes releases the GIL inside the C calls).  lps: list of [T,1025] float32 arrays."""
    from concurrent.futures import ThreadPoolExecutor

    t0 = time.perf_counter()
    done = 0
    for lp in lps:
        orc.predict_logprobs(lp)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    one = done / (time.perf_counter() - t0)
    workers = min(len(lps), os.cpu_count() or 1)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(orc.predict_logprobs, lps))
    allc = len(lps) / (time.perf_counter() - t0)
    return {"post_logits_1_thread_utt_per_s": round(one, 3), "post_logits_all_cores_utt_per_s": round(allc, 3),
            "post_logits_all_cores_threads": workers}


def verse_shaped_logprobs(eng, n: int, T: int, noise: float, boost: float, rng, seed0: int):
    """log-probs synthesised from the token ids of seeded verses (the tests' recipe): n x [T, 1025]"""
    import torch

    from synth import synth_logits

    n_verses = len(eng.tables.s["tok_off"]) // 6
    out = []
    while len(out) < n:
        v = int(rng.integers(0, n_verses))
        ids = eng.tables.token_ids(v, 1).tolist()
        if not (4 <= len(ids) and 2 * len(ids) + 1 <= T):
            continue
        lg = torch.from_numpy(synth_logits(ids, T, seed=seed0 + len(out), noise=noise, boost=boost, rep=2))
        out.append(torch.log_softmax(lg, -1))
    return out


def realistic_mix_leg(eng, audio, lengths, B: int, T: int, steps: int, headline: float):
    """The whole path on a realistic branch mix.  Seeded random weights decode every synthetic clip to a near-empty
    transcript; the reference's published v1 run sent 7 of its 53 clips through the CTC branch (Appendix B of
    SURVEY.md: score < 0.80), the other 46 were settled by the text match.  Here every step still runs the full
    forward on the synthetic clips, but the post-logits stages are fed verse-shaped log-probs at that ratio
    (qv_profile_inject_logprobs): clean ones that pass the 0.80 gate and corrupted ones that fail it and take
    search() + pass 3 + candidate spans + CTC rerank."""
    import numpy as np
    import torch

    rng = np.random.default_rng(20260630)
    n_fail = max(1, round(B * 7 / 53))
    lps = verse_shaped_logprobs(eng, B - n_fail, T, 1.0, 8.0, rng, 5000) + verse_shaped_logprobs(eng, n_fail, T, 3.5, 4.0, rng, 9000)
    order = rng.permutation(B)
    lp = torch.stack([lps[i] for i in order]).cuda(eng.device).contiguous()
    res = eng.decode_retrieve_rerank(lp, [T] * B, want_text=False)
    used = sum(r["use_ctc"] for r in res)
    eng.inject_logprobs(lp, [T] * B)
    try:
        n_ctx = eng.contexts
        inflight = []

        def region(k):
            # same discipline as the headline loop: the rows of the oldest batch in flight are fetched every step
            for _ in range(k):
                inflight.append(eng.predict_batch_async(audio, lengths))
                if len(inflight) >= n_ctx:
                    eng.fetch_results(inflight.pop(0), B, T)
            while inflight:
                eng.fetch_results(inflight.pop(0), B, T)
            torch.cuda.synchronize()

        region(12)
        runs = []
        for _ in range(3):                      # a 130 ms region is at the mercy of one host hiccup: three of them, all reported
            t0 = time.perf_counter()
            region(steps)
            runs.append(time.perf_counter() - t0)
        dt = sorted(runs)[1]                    # the MEDIAN region is the value (round 3 reported the best one)
    finally:
        eng.inject_logprobs(None)
    v = B * steps / dt
    return {"value": round(v, 2), "unit": "utterances/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "runs_utt_per_s": [round(B * steps / r, 1) for r in runs], "value_is": "median of the three regions",
            "gate_failed_utterances_per_batch": used, "batch": B,
            "vs_headline_workload": round(v / headline, 4),
            "what": f"full forward on the synthetic clips + post-logits on verse-shaped log-probs, {B - n_fail} that pass the "
                    f"0.80 text gate : {n_fail} that fail it (the v1 golden run's 46 : 7), same engine and batches in flight "
                    "as the headline line"}


def ingest_leg(eng, audio_np, lengths, B: int, T: int, steps: int, headline: float):
    """The headline loop fed the way a real caller feeds it: every step's batch starts in PINNED HOST memory (8 distinct
    batches), crosses PCIe on a copy stream into one of a ring of device buffers (one step ahead of the engine), and only then
    enters qv_predict_batch_async -- so the host-to-device ingest (640 KB per utterance) is inside the timed region and the
    forward-graph keys rotate with the buffers as they would for a serving loop, instead of one HBM-resident batch replayed
    through one pointer.  Results are fetched as in the headline loop."""
    import numpy as np
    import torch

    n_ctx = eng.contexts
    n_host, n_dev = 8, n_ctx + 2
    host = []
    for k in range(n_host):
        t = torch.from_numpy(np.roll(audio_np, 997 * k, axis=1).copy() if k else audio_np.copy()).pin_memory()
        host.append(t)
    dev = [torch.empty_like(host[0], device=f"cuda:{eng.device}") for _ in range(n_dev)]
    copy_stream = torch.cuda.Stream(device=eng.device)
    ready = [torch.cuda.Event() for _ in range(n_dev)]
    inflight = []

    def upload(i):
        with torch.cuda.stream(copy_stream):
            dev[i % n_dev].copy_(host[i % n_host], non_blocking=True)
            ready[i % n_dev].record(copy_stream)

    def region(k, base):
        upload(base)
        for i in range(base, base + k):
            if i + 1 < base + k:
                upload(i + 1)                      # one step ahead: the copy of the next batch runs under this batch's forward
            torch.cuda.current_stream().wait_event(ready[i % n_dev])
            inflight.append(eng.predict_batch_async(dev[i % n_dev], lengths))
            if len(inflight) >= n_ctx:
                # the buffer of the batch joined here is the next one the ring hands out (n_dev = n_ctx + 2 > batches in flight + the one
                # being uploaded): its audio is never overwritten before its results are on the host
                eng.fetch_results(inflight.pop(0), B, T)
        while inflight:
            eng.fetch_results(inflight.pop(0), B, T)
        torch.cuda.synchronize()

    region(3 * n_dev, 0)                           # every (context, buffer) pair the timed region will see has been captured
    g0 = eng.forward_graph_stats()
    runs = []
    for r in range(3):
        t0 = time.perf_counter()
        region(steps, 1000 * (r + 1))
        runs.append(time.perf_counter() - t0)
    g1 = eng.forward_graph_stats()
    dt = sorted(runs)[1]
    v = B * steps / dt
    return {"value": round(v, 2), "unit": "utterances/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "runs_utt_per_s": [round(B * steps / r, 1) for r in runs], "value_is": "median of the three regions",
            "vs_headline_workload": round(v / headline, 4),
            "h2d_gb_per_s": round(v * audio_np.shape[1] * 4 / 1e9, 2), "host_batches": n_host, "device_buffers": n_dev,
            "forward_graph": {"replays": g1["replays"] - g0["replays"], "captures": g1["captures"] - g0["captures"],
                              "forwards": 3 * steps},
            "what": f"{n_host} distinct pinned host batches -> H2D on a copy stream one step ahead -> ring of {n_dev} device buffers -> "
                    "the same engine and batches in flight as the headline line; rows fetched inside the loop"}


def strong_plan(world: int, rank: int, deal: str, total: int = 2048, seed: int = 20260630):
    """the ragged global batch of --workload strong2048: `total` clip lengths drawn uniformly from 5-30 s, dealt by
    dist.shard_plan; returns (lengths of all clips, order, this rank's original indices longest first)"""
    import numpy as np

    from offline_tarteel_amd import dist as qdist

    rng = np.random.default_rng(seed)
    lengths = rng.integers(80000, 480001, size=total).astype(np.int64)
    order, slices = qdist.shard_plan(lengths, world, deal)
    mine = [int(i) for i in order[slices[rank]]]
    return lengths, order, mine


def strong_pass(eng, base_audio, lengths, mine, per_call: int):
    """one rank's share of the global batch through the engine, `per_call` clips per engine call (longest first, so a call's
    clips have similar lengths), batches in flight over the engine's contexts; returns the result rows in the order of `mine`
    (None for the padding entries).  Clip i is rows i % 64 of the synthetic base batch cut to its length."""
    import torch

    todo = [i for i in mine if i >= 0]
    rows_out = {}
    inflight = []

    def join(t):
        ctx, idx, buf, t_max = t
        for i, r in zip(idx, eng.fetch_results(ctx, len(idx), t_max)):
            rows_out[i] = r

    for s0 in range(0, len(todo), per_call):
        idx = todo[s0: s0 + per_call]
        lens = [int(lengths[i]) for i in idx]
        buf = base_audio[torch.as_tensor([i % base_audio.shape[0] for i in idx], device=base_audio.device), : max(lens)].contiguous()
        for r, n in enumerate(lens):
            buf[r, n:] = 0
        if len(inflight) >= eng.contexts:
            join(inflight.pop(0))
        inflight.append((eng.predict_batch_async(buf, lens), idx, buf, eng.frames_for(max(lens))))
    while inflight:
        join(inflight.pop(0))
    return [rows_out.get(i) for i in mine]


def cpu_baseline(audio_np, n_clips: int, what: str = "10 s clips"):
    """oracle ("port"): fp32 PyTorch-CPU forward + C post-logits, per-file like the reference."""
    import numpy as np
    import torch

    logical, physical = host_cores()

    from oracle import fastconformer_ref as R
    from oracle.oracle import Oracle

    # when onnxruntime and the reference's model file are both on this node, the
    # forward leg is the reference's own call -- InferenceSession.run, batch 1, default session
    # options, CPUExecutionProvider (experiments/c2c-direct-mixed/run.py:48-63) -- and the line
    # is labelled "reference"; otherwise the fp32 PyTorch restatement stands in ("port").
    ort_sess = None
    ref_onnx = os.environ.get("QV_REF_ONNX", "")
    if ref_onnx and os.path.exists(ref_onnx):
        try:
            import onnxruntime as ort

            ort_sess = ort.InferenceSession(ref_onnx, providers=["CPUExecutionProvider"])
        except Exception:
            ort_sess = None
    if ort_sess is not None:
        orc = Oracle()
        n = audio_np.shape[1]
        feed = lambda i: {"audio_signal": audio_np[i: i + 1].astype("float32"), "length": np.array([n], dtype=np.int64)}
        ort_sess.run(None, feed(0))
        t_fwd = t_post = 0.0
        done = 0
        for i in range(n_clips):
            if t_fwd + t_post > 30.0:
                break
            done += 1
            t0 = time.perf_counter()
            lp = ort_sess.run(None, feed(i))[0][0]
            t1 = time.perf_counter()
            orc.predict_logprobs(np.ascontiguousarray(lp, dtype=np.float32))
            t2 = time.perf_counter()
            t_fwd += t1 - t0
            t_post += t2 - t1
        return {
            "value": round(done / (t_fwd + t_post), 4), "unit": "utterances/s", "cores": os.cpu_count(),
            "host_logical_cpus": logical, "host_physical_cores": physical,
            "kind": "reference",
            "sample": f"{done} of the benchmark's {what}, batch 1: onnxruntime CPUExecutionProvider on "
                      f"{os.path.basename(ref_onnx)} with default session options ({t_fwd / done:.2f} s per clip) + C "
                      f"post-logits ({t_post / done:.2f} s per clip)",
        }

    # intra-op threads: batch-1 GEMMs of this size stop scaling (and then collapse) long before a
    # 256-thread host is full; 16 is what the timing below actually uses and reports
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    w = R.random_weights(20260630)
    orc = Oracle()
    n = audio_np.shape[1]
    t_fwd = t_post = 0.0
    # one untimed warm-up clip (mirrors benchmark/runner.py:272-280)
    lp, T = R.forward(w, torch.from_numpy(audio_np[:1]), [n])
    orc.predict_logprobs(lp[0, : int(T[0])].numpy())
    done = 0
    kept = []
    for i in range(n_clips):
        if t_fwd + t_post > 22.0:
            break
        done += 1
        t0 = time.perf_counter()
        lp, T = R.forward(w, torch.from_numpy(audio_np[i: i + 1]), [n])
        t1 = time.perf_counter()
        lpn = np.ascontiguousarray(lp[0, : int(T[0])].numpy())
        orc.predict_logprobs(lpn)
        t2 = time.perf_counter()
        kept.append(lpn)
        t_fwd += t1 - t0
        t_post += t2 - t1
    tot = t_fwd + t_post
    n_clips = done
    # ... and the same batch-1 forward with every physical core as intra-op threads (so that the 16-thread stand-in is
    # not the only defensible choice on record); bounded to ~6 s
    allc = {}
    try:
        used = torch.get_num_threads()
        torch.set_num_threads(max(1, min(physical, os.cpu_count() or 1)))
        R.forward(w, torch.from_numpy(audio_np[:1]), [n])
        t0, k = time.perf_counter(), 0
        while k < min(6, len(audio_np)) and time.perf_counter() - t0 < 6.0:
            R.forward(w, torch.from_numpy(audio_np[k: k + 1]), [n])
            k += 1
        dt_all = time.perf_counter() - t0
        allc = {"forward_all_physical_cores_s_per_clip": round(dt_all / max(k, 1), 4), "forward_all_physical_cores_threads": torch.get_num_threads(),
                "forward_16_threads_s_per_clip": round(t_fwd / max(n_clips, 1), 4)}
        torch.set_num_threads(used)
    except Exception as e:
        allc = {"forward_all_cores_error": f"{type(e).__name__}: {e}"}
    try:   # the post-logits CPU leg single-threaded and on all cores
        split = post_logits_cpu_split(orc, kept * max(1, min(8, (os.cpu_count() or 1) // max(1, len(kept)))))
    except Exception as e:
        split = {"post_logits_split_error": f"{type(e).__name__}: {e}"}
    return {
        "value": round(n_clips / tot, 4), "unit": "utterances/s", "cores": torch.get_num_threads(),
        "host_logical_cpus": logical, "host_physical_cores": physical, **split, **allc,
        "kind": "port",
        "sample": f"{n_clips} of the benchmark's {what}, batch 1 like the reference "
                  f"(fp32 PyTorch forward on {torch.get_num_threads()} threads {t_fwd / n_clips:.2f} s + C post-logits on 1 thread "
                  f"{t_post / n_clips:.2f} s per clip; `cores` = the forward's threads, the host has {physical} physical cores / "
                  f"{logical} logical CPUs); reference ORT/ONNX path unavailable on this node (no onnxruntime, no weight file)",
    }


def extra_legs():
    """Short driver-visible legs of the secondary BASELINE configs, each a fresh `bench.py` process on this GPU (the
    engine's capacity is fixed at creation): configs[2] = 256 clips x 10 s with int4 + int8 weights, once in the
    reference's onnxruntime arithmetic (QV_PREC_ORT_MIXED) and once with f16 operands (QV_PREC_MIXED_INT4_INT8);
    configs[4]'s per-GPU workload = 64 clips x 30 s through the TTA wrapper.  Returns {name: compact line}."""
    import subprocess

    legs = {
        "configs2_b256_ort_mixed": ["--precision", "ort", "--batch", "256", "--steps", "10", "--warmup", "3"],
        "configs2_b256_mixed_f16_operands": ["--precision", "mixed", "--batch", "256", "--steps", "10", "--warmup", "3"],
        "configs4_tta30_per_gpu": ["--workload", "tta30", "--steps", "5", "--warmup", "2"],
        # ... and with the anchor pass gating clips at the reference's ratio instead of every clip (seeded random weights gate all)
        "configs4_tta30_per_gpu_ref_gate_ratio": ["--workload", "tta30", "--tta-mix", "--steps", "8", "--warmup", "3"],
        # north_star quotes clips of 5-30 s: the short end, where a batch of 64 is half the rows of the headline batch, at
        # the headline's batch and at the batch that restores its row count (one call holds up to max_batch clips)
        "short_clips_5s_b64": ["--seconds", "5", "--steps", "30", "--warmup", "8"],
        "short_clips_5s_b128": ["--seconds", "5", "--batch", "128", "--steps", "30", "--warmup", "8"],
        # configs[3]'s global batch as ONE ragged batch of 2,048 clips of 5-30 s through dist.shard_plan (here: one rank takes it all)
        "configs3_strong_ragged_2048": ["--workload", "strong2048", "--steps", "2", "--warmup", "1"],
    }
    out = {}
    for name, flags in legs.items():
        try:
            p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *flags, "--no-cpu-baseline", "--no-post-logits", "--no-extra"],
                               capture_output=True, text=True, timeout=420, cwd=str(ROOT))
            rows = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
            d = json.loads(rows[-1])
            out[name] = {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                         "steps": d["steps"], "dtype": d["dtype"], "workload": d["config"]["workload"],
                         "batches_in_flight": d["config"]["batches_in_flight"],
                         "roofline_kernel": d["roofline"]["kernel"], "roofline_frac": d["roofline"]["frac"],
                         "all_gemm_in_situ_tflops": d["roofline"]["all_gemm_in_situ_tflops"]}
            if "tta_gated_fraction" in d["config"]:
                out[name]["tta_gated_fraction"] = d["config"]["tta_gated_fraction"]
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    return out


def main():
    args = parse()
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a
    # host / library banner when the first communicator is created), so file descriptor 1 points at
    # stderr while the benchmark runs and is restored only for the result line.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dist = None
    # QVERSE_BENCH_FORCE_DIST=1 runs the collective path with a single rank too (1-GPU check of the
    # RCCL plumbing under torch.distributed.run --nproc-per-node 1)
    use_dist = world > 1 or os.environ.get("QVERSE_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)

    import offline_tarteel_amd  # noqa: F401
    from offline_tarteel_amd.engine import Engine
    from synth import synth_audio

    n = int(args.seconds * 16000)
    B = args.batch
    tta = args.workload == "tta30"
    strong = args.workload == "strong2048"
    audio_np = synth_audio(64 if strong else B, n, seed=20260630 + 1000 * rank)
    audio = torch.from_numpy(audio_np).cuda(local_rank).contiguous()
    lengths = [n] * B
    # TTA: the 1.1x-slowed copies are 10 % longer than the clips
    cap = int(n * 1.1) + 1600 if tta else n
    if args.capacity_seconds > 0:
        cap = max(cap, int(args.capacity_seconds * 16000))
    # TTA: anchor pass of the next step + the two perturbed batches of this one: three batches in flight at most
    # TTA: two anchor passes ahead + the two perturbed batches of the step being decided: four batches in flight at most
    n_ctx = min(args.contexts, 4) if tta else args.contexts
    eng = Engine(device=local_rank, with_model=True, seed=20260630, max_batch=B, max_samples=cap,
                 precision={"fp16": 0, "mixed": 1, "ort": 2}[args.precision], skip_unused_passes=not args.literal,
                 contexts=n_ctx)
    # the engine runs fewer batches in flight than asked for when the runtime cannot run its streams side by side
    # (qv_probe_concurrent_streams: GPU_MAX_HW_QUEUES was not in place when HIP initialised)
    n_ctx = eng.contexts
    weights_info = eng.weights_info()   # precision mode + where the quantisation grids came from (qv_weights_info)
    gathered = torch.empty((world * B, 4), dtype=torch.int32, device=f"cuda:{local_rank}") if use_dist else None
    pending = []   # contexts whose packed rows have not been all-gathered yet
    tta_stats = {"gated": 0, "clips": 0}

    # The gathers run on their own stream: predict_batch_async orders a batch's inputs behind everything queued on
    # the caller's stream, and a gather queued there (it waits for an OLDER batch to finish) would hold the next
    # batch back -- measured with the collective path forced on one rank: 12.3 k utt/s on the caller's stream vs
    # the plain run's 17.0 k.
    gstream = torch.cuda.Stream(device=local_rank) if use_dist else None

    def gather(ctx):
        # the path's only exchange: 16 B per utterance, latency-bound
        # host-side join first (the host would block on this context a step later anyway, when it reuses it): nothing
        # queued on the device then waits for an older batch, whichever hardware queue the streams share
        eng.wait(ctx)
        # The rows leave the engine-owned buffer on the CALLER's stream: the batch is finished, so the 1 KB copy waits
        # for nothing, and the context's next batch is ordered behind everything queued on this stream
        # (qv_predict_batch_async records its input event here) -- its k_result cannot overwrite the rows before they
        # have been copied, however far the side stream lags behind a slow peer rank.  Only the collective itself runs
        # on the side stream, behind an event recorded after the copy.
        rows = eng.packed_results(B, ctx)
        copied = torch.cuda.Event()
        copied.record()
        rows.record_stream(gstream)
        with torch.cuda.stream(gstream):
            gstream.wait_event(copied)
            dist.all_gather_into_tensor(gathered, rows)

    t_frames = eng.frames_for(n)
    fetched = {"batches": 0, "rows": 0}

    def fetch(ctx):
        # the path's OUTPUT reaches the host inside the timed region: the qv_result rows of the oldest batch in flight
        # (a 4 KB copy on the context's stream; the host would block on this context a step later anyway, when it reuses it)
        res = eng.fetch_results(ctx, B, t_frames)
        fetched["batches"] += 1
        fetched["rows"] += len(res)

    def step_clips():
        # every step runs the WHOLE hot path on one batch; with contexts > 1 up to that many batches
        # are in flight, so a batch's rows are fetched (1 GPU) / all-gathered (N GPUs) (contexts - 1) steps later
        ctx = eng.predict_batch_async(audio, lengths)
        pending.append(ctx)
        if len(pending) >= n_ctx:
            (gather if use_dist else fetch)(pending.pop(0))

    tta_prev = []   # the previous step's TTA state while its perturbed batches are still in flight
    tta_anchors = []   # contexts of anchor passes launched ahead (four contexts: two ahead)
    tta_lp = None
    if tta and args.tta_mix:
        # anchor passes read verse-shaped log-probs: clean ones (text match >= 0.80, far above the 0.5 gate) and corrupted
        # ones at the v1 golden run's 46 : 7 ratio (they go through search + CTC rerank and come back below 0.5)
        import numpy as np

        rng = np.random.default_rng(20260630)
        n_fail = max(1, round(B * 7 / 53))
        lps = verse_shaped_logprobs(eng, B - n_fail, t_frames, 1.0, 8.0, rng, 5000) + verse_shaped_logprobs(eng, n_fail, t_frames, 3.5, 4.0, rng, 9000)
        tta_lp = torch.stack([lps[i] for i in rng.permutation(B)]).cuda(local_rank).contiguous()

    def anchor_pass():
        if tta_lp is None:
            return eng.predict_batch_async(audio, lengths)
        eng.inject_logprobs(tta_lp, [t_frames] * B)
        try:
            return eng.predict_batch_async(audio, lengths)
        finally:
            eng.inject_logprobs(None)      # the perturbed copies of the gated clips run on their own forward outputs

    def tta_done(st):
        from offline_tarteel_amd import dist as qdist
        from offline_tarteel_amd.plugin import tta_finish

        res = tta_finish(eng, st)
        tta_stats["clips"] += B
        tta_stats["gated"] += sum(1 for r in res if "tta" in r)
        if use_dist:
            with torch.cuda.stream(gstream):
                rows = torch.from_numpy(qdist.pack_results(res)).cuda(local_rank)
                dist.all_gather_into_tensor(gathered, rows)

    def step_tta():
        # c2c-direct-mixed-tta/run.py:117-149 per batch: anchor pass, 0.5 gate (host decision on the
        # fetched scores, as in the reference), 0.9x / 1.1x copies of the gated clips resampled on the GPU
        # and run as further batches, majority / best-score pick; the combined rows are what is gathered.
        # With >= 3 contexts the steps are software-pipelined: this step's anchor pass is launched before the
        # previous step's perturbed batches are joined, so the two run side by side.
        from offline_tarteel_amd.plugin import tta_start

        if n_ctx >= 4:
            # Round 6: TWO anchor passes ahead.  With one (below) the host joins step k - 1's perturbed batches and then waits
            # for anchor k with nothing else on the chip; here anchor k + 1 is already running beside it, and the perturbed
            # batches of step k run beside anchors k + 1 and k + 2 -- what a serving loop with a queue of batches does.
            tta_anchors.append(anchor_pass())
            if len(tta_anchors) >= 2:
                while tta_prev:
                    tta_done(tta_prev.pop(0))
                tta_prev.append(tta_start(eng, audio, lengths, want_text=False, anchor_ctx=tta_anchors.pop(0)))
        elif n_ctx >= 3:
            ctx = anchor_pass()
            if tta_prev:
                tta_done(tta_prev.pop())
            tta_prev.append(tta_start(eng, audio, lengths, want_text=False, anchor_ctx=ctx))
        else:
            ctx = anchor_pass()
            tta_done(tta_start(eng, audio, lengths, want_text=False, anchor_ctx=ctx))

    s_lengths = s_order = s_mine = None
    if strong:
        from offline_tarteel_amd import dist as qdist

        s_lengths, s_order, s_mine = strong_plan(world, rank, args.deal)

    def step_strong():
        # the WHOLE global batch: this rank's share through the engine, then the path's one exchange (16 B per utterance)
        rows = strong_pass(eng, audio, s_lengths, s_mine, B)
        if use_dist:
            full = qdist.all_gather_results(torch.from_numpy(qdist.pack_results(rows)).cuda(local_rank), s_order, len(s_lengths))
            assert full.shape == (len(s_lengths), 4)

    step = step_tta if tta else step_strong if strong else step_clips

    def sync_all():
        while tta_anchors:       # the anchors launched ahead are decided too: every timed step's clips come back inside the region
            from offline_tarteel_amd.plugin import tta_start

            while tta_prev:
                tta_done(tta_prev.pop(0))
            tta_prev.append(tta_start(eng, audio, lengths, want_text=False, anchor_ctx=tta_anchors.pop(0)))
        while tta_prev:
            tta_done(tta_prev.pop())
        while pending:
            (gather if use_dist else fetch)(pending.pop(0))
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    fetched["batches"] = fetched["rows"] = 0
    import gc

    gc.collect()
    gc.disable()          # the timed region is tens of milliseconds of host-driven launches: no collector pause inside it
    graph0 = eng.forward_graph_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    gc.enable()
    graph1 = eng.forward_graph_stats()
    # how the timed region's forwards were issued: as hipGraph replays (multi-context engines, shape seen before -- the
    # warm-up captures it) or as ~300 plain launches each (single-context engines, QVERSE_FWD_GRAPH=0)
    forward_graph = {"replays_in_timed_region": graph1["replays"] - graph0["replays"],
                     "captures_in_timed_region": graph1["captures"] - graph0["captures"], "captures_before": graph0["captures"]}
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = (len(s_lengths) if strong else world * B) * args.steps / dt
    if not tta and not strong and not use_dist:
        assert fetched["batches"] == args.steps and fetched["rows"] == B * args.steps, fetched   # every timed batch came back

    # The driver's timed region is K = 20 steps (~73 ms): the same loop over a region two orders of magnitude longer, in
    # the same process on the same engine, so that the short region's number is corroborated in the same output line
    # (default line only: one GPU, configs[1]).
    long_region = None
    if (world == 1 and not tta and not strong and not use_dist and args.precision == "fp16" and B == 64 and args.seconds == 10.0
            and not args.no_extra):
        n_long = 2000
        fetched["batches"] = fetched["rows"] = 0
        sync_all()
        t1 = time.perf_counter()
        for _ in range(n_long):
            step()
        sync_all()
        dt_long = time.perf_counter() - t1
        assert fetched["batches"] == n_long, fetched
        long_region = {"steps": n_long, "ms_per_step": round(dt_long / n_long * 1e3, 3), "value": round(B * n_long / dt_long, 2),
                       "unit": "utterances/s", "note": "same step() as the timed region, run after it"}

    # sanity: results come back and look like predictions
    if strong:
        audio, lengths = audio[: min(B, 64)].contiguous(), [n] * min(B, 64)
        B = len(lengths)
    res = eng.predict_batch(audio, lengths, want_text=False)
    if not os.environ.get("QVERSE_SKIP"):   # (timing experiments drop kernels: results are meaningless then)
        assert len(res) == B and all(r["t_frames"] == eng.frames_for(n) for r in res)
    used_ctc = sum(r["use_ctc"] for r in res)
    rows_per_launch = B * eng.frames_for(n)   # M of the encoder GEMMs

    # ---- roofline of the dominant kernel -----------------------------------------------------
    # Dominant kernel by time (profiles/*_kernel_stats.csv): the GEMM family; its single-shape
    # member with the largest share is the FFN-up GEMM ([B*T,512] x [512,2048] + Swish, 2 per layer:
    # k_gemm256<f16_swish> on 256 x 256 tiles when the grid has >= 160 of them, else k_gemm<f16_swish,128>).  It is timed live with HIP events on the launch
    # stream over a back-to-back replay on the engine's own buffers; the in-situ, per-launch event
    # timing of EVERY GEMM class of a few full steps is reported next to it (that one includes the
    # event/launch gap of each launch, so it reads lower).
    roof = None
    if rank == 0:
        # Dominant kernel by time in rocprofv3's table of this very command (profiles/r0*_kernel_stats.csv): the
        # RESIDUAL-epilogue GEMM class -- FFN-down [M,2048]x[2048,512] (2 per layer) and the attention out-projection /
        # second pointwise convolution [M,512]x[512,512] (1 + 1 per layer), one kernel name, 68 launches per step.  It is
        # timed live with HIP events on the launch stream over back-to-back replays of both shapes on the engine's own
        # buffers (alpha = 0: the residual stream is left alone) and reported launch-weighted, as rocprofv3 averages it.
        # The FFN-up GEMM ([M,512]x[512,2048] + Swish, the single shape with the most FLOPs) is kept next to it, and the
        # in-situ, per-launch event timing of EVERY GEMM class of a few full steps (that one includes the event / launch
        # gap of each launch, so it reads lower).
        n_cu = 256
        r_dn, r_out, rep = eng.replay_gemm(1, iters=60), eng.replay_gemm(3, iters=60), eng.replay_gemm(0, iters=100)
        cls_flops = 34 * r_dn["flops"] + 34 * r_out["flops"]
        cls_us = 34 * r_dn["avg_us"] + 34 * r_out["avg_us"]
        ach_cls = cls_flops / (cls_us * 1e-6) / 1e12
        ach = rep["flops"] / (rep["avg_us"] * 1e-6) / 1e12

        def tiles_of(r, n_cols):   # blocks of that launch: 256 x 256 tiles or 128 x 128 / 128 x 64 ones
            if r["kernel"].startswith("k_gemm256"):
                return ((rows_per_launch + 255) // 256) * (n_cols // 256)
            bn = 64 if ",64" in r["kernel"] else 128
            return ((rows_per_launch + 127) // 128) * (n_cols // bn)

        cus_dn = min(n_cu, tiles_of(r_dn, 512))
        # HBM traffic / MFMA utilisation are NOT measured in this run (PMC counters need rocprofv3): they are quoted from
        # the newest committed counter summary, and only for the kernel and shape that summary was taken at
        def quoted(pattern, key, kernel):
            try:
                f = sorted((ROOT / "profiles").glob(pattern))[-1]
                doc = json.loads(f.read_text())
                if int(doc.get("rows", 0)) == rows_per_launch and args.precision == "fp16":
                    return doc["kernels"].get(kernel, {}).get(key), f"profiles/{f.name} (rocprofv3 --pmc over tools/gemm_bench, same kernel, M = {rows_per_launch}; not collected in this run)"
                return None, f"no committed counter summary for M = {rows_per_launch}, weights {args.precision}"
            except Exception:
                return None, None

        traffic, traffic_source = quoted("r*_pmc_traffic.json", "hbm_bytes_per_launch", r_dn["kernel"])
        mfma_util, mfma_source = quoted("r*_mfma_busy.json", "mfma_util", r_dn["kernel"])
        insitu, insitu_source = None, None
        try:
            for f in sorted((ROOT / "profiles").glob("r*_mfma_in_situ*.json"), reverse=True):
                doc = json.loads(f.read_text())
                if int(doc.get("rows", 0)) == rows_per_launch and doc.get("weights") == args.precision:
                    insitu = doc["all_gemm"]["mfma_util"]
                    insitu_source = (f"profiles/{f.name} (rocprofv3 --pmc over THIS command with --contexts 1, weighted over all "
                                     f"{doc['all_gemm']['launches']} GEMM launches; not collected in this run)")
                    break
        except Exception:
            pass
        up_traffic, _ = quoted("r*_pmc_traffic.json", "hbm_bytes_per_launch", rep["kernel"])
        up_util, _ = quoted("r*_mfma_busy.json", "mfma_util", rep["kernel"])
        eng.profile_gemm(True)
        nprof = max(2, min(args.steps, 5))
        for _ in range(nprof):
            eng.predict_batch_async(audio, lengths)
            torch.cuda.synchronize()   # one batch at a time here: per-launch timings must not overlap
        classes = eng.profile_gemm_read()
        eng.profile_gemm(False)
        gemm_ms = sum(c["ms"] for c in classes)
        tf = lambda r: round(r["flops"] / (r["avg_us"] * 1e-6) / 1e12, 1)  # noqa: E731
        roof = {
            "bound": "mfma", "kernel": r_dn["kernel"],
            "shape": "residual-epilogue GEMMs, launch-weighted as in the kernel table: 34 x FFN-down [M,2048]x[2048,512] + 34 x "
                     "attention-out / pointwise-conv-2 [M,512]x[512,512] per step",
            "achieved": round(ach_cls, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach_cls / PEAK_F16_TFLOPS, 4),
            "traffic": traffic, "traffic_source": traffic_source, "mfma_util_pmc": mfma_util, "mfma_util_source": mfma_source,
            "mfma_util_in_situ": insitu, "mfma_util_in_situ_source": insitu_source,
            "flops_per_launch": cls_flops / 68, "avg_launch_us": round(cls_us / 68, 2), "launches": 120,
            "members": {"ffn_down": {"kernel": r_dn["kernel"], "tflops": tf(r_dn), "avg_launch_us": round(r_dn["avg_us"], 2),
                                     "blocks": tiles_of(r_dn, 512), "cus_occupied": cus_dn,
                                     "frac_of_occupied_cus_peak": round(tf(r_dn) / (PEAK_F16_TFLOPS * cus_dn / n_cu), 4)},
                        "out_proj_pw2": {"kernel": r_out["kernel"], "tflops": tf(r_out), "avg_launch_us": round(r_out["avg_us"], 2),
                                         "blocks": tiles_of(r_out, 512)}},
            "note": ("with >= 3 batches in flight the N = 512 GEMMs run as 64 tiles of 256 x 256 on 64 of the 256 CUs (fewest "
                     "CU-microseconds per GEMM; the other batches' kernels take the idle CUs): `frac` prices them against the "
                     "WHOLE chip's peak, `frac_of_occupied_cus_peak` against the CUs they hold") if cus_dn < n_cu else
                    "one batch at a time: 128-wide tiles, 252 blocks",
            "ffn_up": {"kernel": rep["kernel"], "shape": rep["shape"], "achieved": round(ach, 2), "frac": round(ach / PEAK_F16_TFLOPS, 4),
                       "avg_launch_us": round(rep["avg_us"], 2), "flops_per_launch": rep["flops"], "traffic": up_traffic,
                       "mfma_util_pmc": up_util},
            "tile_policy": (f"{n_ctx} batches in flight: 256 x 256 tiles (one block per CU) for every GEMM with N % 256 == 0"
                            + ("" if n_ctx >= 4 else
                               " and >= 128 such tiles, or K >= 2048" if n_ctx >= 3 else " and >= 160 such tiles")
                            + " (with several batches in flight the small-grid GEMMs run on 64-128 CUs: fewer CU-microseconds per GEMM, the "
                              "other batches' kernels take the idle CUs; profiles/archive/r02_f_tile_policy_sweep.txt), 128-wide tiles otherwise; "
                              "other_gemms are stand-alone replays under that policy"),
            "other_gemms": {eng.REPLAY_SHAPES[w]: tf(eng.replay_gemm(w, 50)) for w in (2, 4)},
            "all_gemm_in_situ_tflops": round(sum(c["flops"] for c in classes) / (gemm_ms * 1e-3) / 1e12, 2),
            "all_gemm_in_situ_ms_per_step": round(gemm_ms / nprof, 3),
            "all_gemm_in_situ_note": ("per-launch HIP-event times of every GEMM, taken ONE BATCH AT A TIME (a device synchronise after "
                                      "each batch, so that event brackets of different batches cannot overlap); their sum may exceed "
                                      "ms_per_step of the timed loop, where up to `batches_in_flight` batches share the chip and small-grid "
                                      "GEMMs of one batch run beside kernels of another"),
            "end_to_end_frac": round(value / world * FLOP_PER_UTT_10S * (args.seconds / 10.0) * (3.0 if tta else 1.0) / 1e12 / PEAK_F16_TFLOPS, 5),
        }

    post = None
    mix = None
    ingest = None
    if rank == 0 and not args.no_post_logits and not tta and not strong and world == 1:
        try:
            ingest = ingest_leg(eng, audio_np, lengths, B, eng.frames_for(n), max(10, min(args.steps, 30)), value)
        except Exception as e:
            ingest = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and not args.no_post_logits:
        try:
            post = post_logits_legs(eng, min(B, 64), min(126, eng.frames_for(cap)))
        except Exception as e:
            post = {"error": f"{type(e).__name__}: {e}"}
        if not tta and not strong and world == 1:
            try:
                mix = realistic_mix_leg(eng, audio, lengths, B, eng.frames_for(n), max(10, min(args.steps, 30)), value)
            except Exception as e:
                mix = {"error": f"{type(e).__name__}: {e}"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(audio_np, args.cpu_sample,
                               f"{args.seconds:g} s clips" + (" (anchor pass only, no TTA copies)" if tta else ""))
        except Exception as e:  # the baseline leg must never take the GPU number down with it
            cpu = {"value": None, "unit": "utterances/s", "cores": os.cpu_count(), "kind": "port",
                   "sample": f"failed: {type(e).__name__}: {e}"}

    extra = None
    default_line = (world == 1 and not tta and not strong and args.precision == "fp16" and B == 64 and args.seconds == 10.0)
    if rank == 0 and default_line and not args.no_extra:
        eng.close()   # the legs create engines of their own on this GPU
        extra = extra_legs()

    if rank == 0:
        mixed = args.precision != "fp16"
        wdesc = ("fp16 weights" if not mixed else
                 "int4 (block-128) Linear + int8 (per-channel) pointwise-conv weights dequantised in the MFMA operand fetch, f16 activations"
                 if args.precision == "mixed" else
                 "the reference's onnxruntime arithmetic: int4 (block-128) Linear weights on f16 activations; every Conv as "
                 "DynamicQuantizeLinear (per-utterance uint8 activations) -> ConvInteger (per-tensor int8 weights, int32 "
                 "accumulation: i8 MFMA / exact integer stencils) -> float32 rescale")
        if strong:
            secs = float(s_lengths.sum()) / 16000.0
            workload = (f"c2c-direct-mixed hot path, ONE ragged global batch of {len(s_lengths)} synthetic clips of 5-30 s ({secs:.0f} audio-seconds) per "
                        f"step, dealt to {world} rank(s) by dist.shard_plan ({args.deal}), {B} clips per engine call, rows back through "
                        f"dist.all_gather_results, {wdesc} (BASELINE.json configs[3]'s global batch as a STRONG-scaling workload); seeded random weights")
        elif tta:
            cfg_name = "BASELINE.json configs[4]" + ("" if world == 8 else f" workload on {world} GPU(s)")
            workload = (f"c2c-direct-mixed-tta hot path, batch={B}/GPU synthetic {args.seconds:g} s/16 kHz clips: anchor pass, "
                        "0.5 confidence gate, GPU 0.9x/1.1x speed-perturbed copies of the gated clips, majority / best-score "
                        f"pick, {wdesc} ({cfg_name}); " +
                        ("anchor passes read verse-shaped log-probs at the v1 golden run's 46 : 7 branch ratio (--tta-mix), the forward runs on every clip"
                         if args.tta_mix else "seeded random weights (real ONNX absent) gate every clip"))
        else:
            if not mixed and B == 64 and args.seconds == 10.0 and world < 8:
                cfg_name = "BASELINE.json configs[1]" + ("" if world == 1 else f" per GPU, {world} GPUs")
            elif mixed and B == 256 and world == 1:
                cfg_name = "BASELINE.json configs[2]"
            elif B == 256 and world == 8 and args.seconds == 10.0:
                cfg_name = "BASELINE.json configs[3]: global batch 2048"
            else:
                cfg_name = "not a BASELINE.json configuration"
            workload = (f"c2c-direct-mixed hot path, batch={B}/GPU synthetic {args.seconds:g} s/16 kHz clips, "
                        f"FastConformer-CTC forward ({wdesc}) + greedy decode + verse retrieval + CTC rerank "
                        f"({cfg_name}); seeded random weights (real ONNX absent)")
        out = {
            "metric": ("utterances/sec (5-30 s @16 kHz, ragged global batch of 2048)" if strong else
                       f"utterances/sec ({args.seconds:g} s @16 kHz{', TTA 0.9x/1.0x/1.1x' if tta else ''})"),
            "value": round(value, 2), "unit": "utterances/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f16" if not mixed else "f16 (int4/int8 weights)" if args.precision == "mixed" else "f16 Linear (int4 weights) + u8 x i8 -> i32 Conv",
            "data": "synthetic",
            "config": {"workload": workload,
                       "global_batch": len(s_lengths) if strong else world * B, "seconds": args.seconds, "parallelism": f"dp{world}",
                       "gate_failed_utterances_per_batch": used_ctc,
                       "skip_unused_passes": not args.literal, "weights": args.precision, "weights_effective": weights_info,
                       "batches_in_flight": n_ctx, "forward_graph": forward_graph, "engine_capacity_seconds": round(cap / 16000.0, 2),
                       "concurrent_streams_probe": int(eng.lib.qv_probe_concurrent_streams())},
            "roofline": roof, "cpu_baseline": cpu, "post_logits": post, "realistic_mix": mix, "ingest": ingest, "long_region": long_region, "extra": extra,
        }
        if strong:
            out["audio_seconds_per_s"] = round(float(s_lengths.sum()) / 16000.0 * args.steps / dt, 1)
            out["config"]["deal"] = args.deal
            out["config"]["rank0_share_audio_seconds"] = round(float(sum(s_lengths[i] for i in s_mine if i >= 0)) / 16000.0, 1)
        if tta:
            out["config"]["tta_gated_fraction"] = round(tta_stats["gated"] / max(1, tta_stats["clips"]), 3)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    eng.close()
    if use_dist:
        # the last gathered block must hold this rank's own rows
        own = gathered[rank * B: (rank + 1) * B].cpu()
        assert own.shape == (B, 4)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
