"""CPU tests of the host-side mirror: runner scoring, emissions, plugin surface, audio ingest,
data-parallel sharding + all-gather over gloo (world size 2)."""

import json
import os
import struct
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_runner_scoring_matches_reference_known_answers(golden_dir):
    from offline_tarteel_amd.benchmark.runner import predict_to_emissions, score_sequence

    sc = json.loads((golden_dir / "scoring_cases.json").read_text(encoding="utf-8"))
    for c in sc["score_sequence"]:
        assert score_sequence(c["expected"], c["predicted"]) == c["out"]
    for c in sc["emissions"]:
        assert predict_to_emissions(c["in"]) == c["out"]


def test_plugins_load_by_path_and_expose_contract():
    from offline_tarteel_amd.benchmark.runner import EXPERIMENT_REGISTRY, load_module

    for name, path in EXPERIMENT_REGISTRY.items():
        mod = load_module(name.replace("-", "_"), path)
        for fn in ("predict", "transcribe", "model_size"):
            assert callable(getattr(mod, fn)), (name, fn)
        assert mod.model_size() == 0  # no weight file configured


def test_missing_model_is_file_not_found(monkeypatch, tmp_path):
    from offline_tarteel_amd import plugin

    monkeypatch.delenv("QVERSE_WEIGHTS", raising=False)
    monkeypatch.delenv("QVERSE_RANDOM_WEIGHTS", raising=False)
    monkeypatch.setattr(plugin, "_engine", None)
    wav = tmp_path / "a.wav"
    _write_wav(wav, np.zeros(1600, np.int16), 16000)
    with pytest.raises(FileNotFoundError):
        plugin.predict(str(wav))


def test_runner_records_empty_prediction_on_plugin_error(tmp_path, monkeypatch):
    """missing audio -> skipped; exception in predict -> empty emissions, latency 0.0."""
    from offline_tarteel_amd.benchmark import runner

    corpus = tmp_path / "corpus"
    corpus.mkdir()
    _write_wav(corpus / "x.wav", np.zeros(3200, np.int16), 16000)
    (corpus / "manifest.json").write_text(json.dumps({"samples": [
        {"id": "present", "file": "x.wav", "surah": 1, "ayah": 1, "category": "short"},
        {"id": "absent", "file": "nope.wav", "surah": 1, "ayah": 2, "category": "short"}]}))
    monkeypatch.delenv("QVERSE_WEIGHTS", raising=False)
    monkeypatch.delenv("QVERSE_RANDOM_WEIGHTS", raising=False)
    exp = runner.discover_experiments("c2c-direct-mixed")[0]
    res = runner.run_experiment(exp, runner.load_manifest(corpus), corpus)
    assert res["total"] == 1 and res["per_sample"][0]["id"] == "present"
    assert res["per_sample"][0]["predicted"] == [] and res["per_sample"][0]["latency"] == 0.0
    assert res["recall"] == 0.0
    p = runner.save_results([res], results_dir=tmp_path / "results")
    latest = json.loads((tmp_path / "results" / "latest.json").read_text())
    assert latest[0]["name"] == "c2c-direct-mixed" and latest[0]["source_file"] == p.name


def test_runner_batch_group_failure_falls_back_to_per_file(tmp_path):
    """--batch N: one bad file makes predict_batch raise for the whole group; the runner then retries
    file by file so that only the offending sample is recorded as empty (the reference isolates failures
    per sample, runner.py:297-325)."""
    from offline_tarteel_amd.benchmark import runner

    corpus = tmp_path / "corpus"
    corpus.mkdir()
    samples = [{"id": f"s{i}", "file": f"s{i}.wav", "surah": 1, "ayah": i + 1, "category": "short"} for i in range(3)]
    for s in samples:
        (corpus / s["file"]).write_bytes(b"RIFF")
    (corpus / "manifest.json").write_text(json.dumps({"samples": samples}))
    plug = tmp_path / "run.py"
    plug.write_text(
        "def predict(p):\n"
        "    if p.endswith('s1.wav'): raise ValueError('undecodable')\n"
        "    a = int(p[-5]) + 1\n"
        "    return {'surah': 1, 'ayah': a, 'ayah_end': a, 'score': 1.0}\n"
        "def predict_batch(ps):\n"
        "    return [predict(p) for p in ps]\n"
        "def model_size():\n    return 0\n")
    res = runner.run_experiment({"name": "grp", "run_path": plug, "model_name": None}, samples, corpus, batch=3)
    got = {r["id"]: r for r in res["per_sample"]}
    assert got["s0"]["recall"] == 1.0 and got["s2"]["recall"] == 1.0
    assert got["s1"]["predicted"] == [] and got["s1"]["latency"] == 0.0
    assert abs(res["recall"] - 2 / 3) < 1e-12


def test_runner_transcribe_only_experiment_both_modes(tmp_path, oracle):
    """An experiment without predict() goes through StreamingPipeline, as in the reference's runner
    (runner.py:309-321): chunked in streaming mode, run_on_full_transcript otherwise; the mode shows
    in the result name and in latest.json's chunk_seconds.  The matching steps are the CPU oracle's
    here (host logic only; the HIP steps are checked in tests/test_gpu_tracker.py)."""
    from offline_tarteel_amd.benchmark import runner
    from offline_tarteel_amd.streaming import StreamingPipeline
    from oracle.tracker_ref import MatchVerseOracle, TrackerOracle
    from test_oracle_tracker import oracle_matcher

    text = oracle.verse_text(oracle.verse_index(112, 1))
    corpus = tmp_path / "corpus"
    corpus.mkdir()
    _write_wav(corpus / "x.wav", np.zeros(48000, np.int16), 16000)
    (corpus / "manifest.json").write_text(json.dumps({"samples": [
        {"id": "s", "file": "x.wav", "surah": 112, "ayah": 1, "category": "short"}]}))
    plug = tmp_path / "run.py"
    plug.write_text(f"def transcribe(audio_path):\n    return {text!r}\n\ndef model_size():\n    return 7\n", encoding="utf-8")
    mv = MatchVerseOracle(oracle)
    pipe = StreamingPipeline(matcher=oracle_matcher(TrackerOracle(oracle)),
                             match_verse_fn=lambda t, max_span, hint: mv.match_verse(t, max_span=max_span, hint=hint))
    exp = {"name": "mock-asr", "run_path": plug, "model_name": None}
    samples = runner.load_manifest(corpus)
    full = runner.run_experiment(exp, samples, corpus, pipeline=pipe)
    assert full["name"] == "mock-asr" and full["model_size"] == 7
    assert [(e["surah"], e["ayah"]) for e in full["per_sample"][0]["predicted"]] == [(112, 1)] and full["recall"] == 1.0
    st = runner.run_experiment(exp, samples, corpus, mode="streaming", chunk_seconds=3.0, pipeline=pipe)
    assert st["name"] == "mock-asr (stream 3s)"
    assert [(e["surah"], e["ayah"]) for e in st["per_sample"][0]["predicted"]] == [(112, 1)]
    runner.save_results([st], mode="streaming", results_dir=tmp_path / "results", chunk_seconds=3.0)
    latest = json.loads((tmp_path / "results" / "latest.json").read_text())
    assert latest[0]["mode"] == "streaming" and latest[0]["chunk_seconds"] == 3.0


def _write_wav(path, pcm16, sr, channels=1):
    data = pcm16.astype("<i2").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack(
        "<IHHIIHH", 16, 1, channels, sr, sr * 2 * channels, 2 * channels, 16) + b"data" + struct.pack("<I", len(data))
    Path(path).write_bytes(hdr + data)


def test_load_audio_wav_mono_and_resample(tmp_path):
    from offline_tarteel_amd.audio import load_audio, speed_perturb

    t = np.arange(16000) / 16000.0
    x = (np.sin(2 * np.pi * 440 * t) * 12000).astype(np.int16)
    _write_wav(tmp_path / "m.wav", x, 16000)
    a = load_audio(str(tmp_path / "m.wav"))
    assert a.dtype == np.float32 and len(a) == 16000 and np.allclose(a, x / 32768.0)
    st = np.stack([x, -x], axis=1).reshape(-1)
    _write_wav(tmp_path / "s.wav", st, 44100, channels=2)
    b = load_audio(str(tmp_path / "s.wav"))
    assert abs(len(b) - round(16000 * 16000 / 44100)) <= 1 and np.abs(b).max() < 1e-3  # L/R cancel
    from scipy.signal import resample_poly

    assert np.array_equal(speed_perturb(a, 0.9), resample_poly(a, 9, 10).astype("float32"))
    assert speed_perturb(a, 1.0) is a
    with pytest.raises(ValueError):
        (tmp_path / "c.mp3").write_bytes(b"ID3\x00" * 8)
        load_audio(str(tmp_path / "c.mp3"))


def upfirdn_restatement(x, taps, up, down, m0, n_out):
    """numpy restatement of k_upfirdn's arithmetic (include/qverse.h: qv_upfirdn): per output
    sample, products accumulated in float32 in ascending input order."""
    P = (len(taps) + up - 1) // up
    hp = np.zeros(up * P, np.float32)
    hp[: len(taps)] = taps
    m = np.arange(m0, m0 + n_out, dtype=np.int64)
    xi, t = (m * down) // up, (m * down) % up
    acc = np.zeros(n_out, np.float32)
    for j in range(P):
        i = xi - (P - 1) + j
        ok = (i >= 0) & (i < len(x))
        prod = (x[np.clip(i, 0, len(x) - 1)] * hp[t + up * (P - 1 - j)]).astype(np.float32)
        acc = np.where(ok, (acc + prod).astype(np.float32), acc)
    return acc


def test_resample_plan_and_fir_order_match_scipy_bitwise():
    """a15: the host plan (taps, first kept sample, output count) + the kernel's summation order
    reproduce scipy.signal.resample_poly on float32 input bit for bit."""
    from scipy.signal import resample_poly

    from offline_tarteel_amd.audio import resample_plan

    rng = np.random.default_rng(3)
    for n_in, (up, down) in ((16000, (9, 10)), (16001, (11, 10)), (777, (9, 10)), (5, (11, 10)), (4410, (160, 441)),
                             (1000, (18, 20))):
        x = rng.standard_normal(n_in).astype(np.float32)
        u, d, taps, m0, n_out = resample_plan(up, down, n_in)
        want = resample_poly(x, up, down)
        assert want.dtype == np.float32 and len(want) == n_out
        got = upfirdn_restatement(x, taps, u, d, m0, n_out)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (n_in, up, down)


def test_shard_plan_covers_batch_once():
    from offline_tarteel_amd.dist import shard_plan

    lens = [5, 30, 12, 7, 22, 9, 18]
    order, slices = shard_plan(lens, 4)
    assert sorted(i for i in order if i >= 0) == list(range(7)) and len(order) == 8
    assert [order[s].tolist() for s in slices][0][0] == 1  # longest first


_GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import offline_tarteel_amd
from offline_tarteel_amd.dist import shard_plan, pack_results, all_gather_results, unpack_results
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lens = [100, 400, 250, 50, 320]
order, slices = shard_plan(lens, world)
mine = order[slices[rank]]
# stand-in for the engine: every utterance "predicts" (surah=idx+1, ayah=len%7+1)
res = [{"surah": (int(i) + 1 if i >= 0 else 0), "ayah": (lens[i] % 7 + 1 if i >= 0 else 0), "ayah_end": None,
        "score": (lens[i] / 1000.0 if i >= 0 else 0.0)} for i in mine]
full = all_gather_results(torch.from_numpy(pack_results(res)), order, len(lens))
got = unpack_results(full)
assert [g["surah"] for g in got] == [1, 2, 3, 4, 5], got
assert all(abs(g["score"] - l / 1000.0) < 1e-6 for g, l in zip(got, lens))
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_all_gather_results_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script), str(ROOT)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2



_GLOO_TTA_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import offline_tarteel_amd
from offline_tarteel_amd.dist import pack_results, unpack_results
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
B = 6
gathered = torch.empty((world * B, 4), dtype=torch.int32)
# what bench.py's tta_done() gathers per step (configs[4]): the rows of the TTA decision rule's result dicts -- clips the
# gate let through keep their anchor result, gated clips carry the majority / best-score pick and a "tta" record, spans
# have an ayah_end, an unrecognised clip is all zeros with ayah_end None
def tta_results(r, step):
    out = []
    for i in range(B):
        k = r * 100 + step * 10 + i
        d = {"surah": k % 114 + 1, "ayah": k % 7 + 1, "ayah_end": (k % 7 + 3 if i % 3 == 0 else None), "score": 0.25 + 0.001 * k}
        if i % 2:
            d["tta"] = {"speeds": [0.9, 1.0, 1.1], "votes": 2}
        if i == B - 1:
            d = {"surah": 0, "ayah": 0, "ayah_end": None, "score": 0.0}
        out.append(d)
    return out
for step in range(3):      # one gather per step, as in the bench loop
    rows = torch.from_numpy(pack_results(tta_results(rank, step)))
    dist.all_gather_into_tensor(gathered, rows)
    got = unpack_results(gathered.numpy())
    want = [d for r in range(world) for d in tta_results(r, step)]
    assert len(got) == world * B
    for g, w in zip(got, want):
        assert (g["surah"], g["ayah"]) == (w["surah"], w["ayah"]), (g, w)
        assert g["ayah_end"] == ((w["ayah_end"] or w["ayah"]) if w["surah"] else None), (g, w)
        assert abs(g["score"] - np.float32(w["score"])) == 0.0, (g, w)
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_tta_rows_all_gather_gloo_world2(tmp_path):
    """configs[4]'s exchange (bench.py tta_done): the decision rule's combined rows, packed on the host, one all-gather per
    step -- world size 2 on gloo; every rank must see every rank's rows in rank order, scores bit-exact."""
    script = tmp_path / "w.py"
    script.write_text(_GLOO_TTA_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", str(script), str(ROOT)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


# ---------------------------------------------------------------------------------------------------------------------
# the multi-GPU PRODUCT entry: python -m torch.distributed.run ... -m offline_tarteel_amd.benchmark.runner
_STUB_PLUGIN = r'''
"""A predict()/predict_batch()/model_size() plugin that needs no GPU: the "prediction" is a function of the file's
sample count, so a sharded run must reproduce a single-process run row for row."""
import struct
from pathlib import Path


def _n(path):
    data = Path(path).read_bytes()
    return (len(data) - 44) // 2


def predict(audio_path):
    if "bad" in Path(audio_path).name:
        raise ValueError("undecodable file")
    n = _n(audio_path)
    if (n // 100) % 5 == 0:
        return {"surah": 0, "ayah": 0, "ayah_end": None, "score": 0.0, "transcript": "", "candidates": []}
    k = n // 100
    return {"surah": k % 114 + 1, "ayah": k % 7 + 1, "ayah_end": (k % 7 + 1 + k % 3), "score": round((k * 7919 % 10007) / 10007, 4),
            "transcript": "x", "source": "text"}


def predict_batch(paths):
    if any("bad" in Path(p).name for p in paths):
        raise ValueError("one undecodable file in the batch")
    return [predict(p) for p in paths]


def model_size():
    return 1234
'''


def _write_wav_n(path, n):
    import struct

    pcm = (np.arange(n) % 251).astype("<i2").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16)
    path.write_bytes(hdr + b"data" + struct.pack("<I", len(pcm)) + pcm)


def _stub_corpus(tmp_path):
    corpus = tmp_path / "corpus"
    corpus.mkdir()
    exps = tmp_path / "exps" / "stub-exp"
    exps.mkdir(parents=True)
    (exps / "run.py").write_text(_STUB_PLUGIN)
    lens = [3100, 900, 12300, 700, 4500, 8100, 2300, 15000, 5200, 6100, 1900]
    samples = []
    for i, n in enumerate(lens):
        name = f"clip_{i:02d}.wav" if i != 4 else "bad_04.wav"
        _write_wav_n(corpus / name, n)
        k = n // 100
        samples.append({"id": f"s{i}", "file": name, "surah": k % 114 + 1, "ayah": k % 7 + 1, "category": "short" if n < 5000 else "long"})
    samples.append({"id": "missing", "file": "not_there.wav", "surah": 1, "ayah": 1, "category": "short"})
    samples[2]["expected_verses"] = [{"surah": samples[2]["surah"], "ayah": samples[2]["ayah"]},
                                     {"surah": samples[2]["surah"], "ayah": samples[2]["ayah"] + 1}]
    (corpus / "manifest.json").write_text(json.dumps({"samples": samples}))
    return corpus, tmp_path / "exps"


def _strip(results):
    out = json.loads(json.dumps(results))
    for r in out:
        r.pop("avg_latency", None)
        r.pop("world_size", None)
        for row in r["per_sample"]:
            row.pop("latency", None)
            row.pop("error", None)
    return out


@pytest.mark.parametrize("deal", ["strided", "contiguous"])
def test_sharded_runner_entry_gloo_world2_equals_single_process(tmp_path, deal):
    """The product's multi-GPU entry on two gloo ranks with a stub plugin: rank 0 reads the manifest, shard_plan deals the
    length-sorted files, both ranks predict (one batch holds an undecodable file: retried file by file, as in the
    single-process runner), all_gather_results restores the manifest order, rank 0 scores and writes results/*.json --
    row for row what the single-process runner writes (scores included: 4-decimal values survive the float32 rows)."""
    corpus, exps = _stub_corpus(tmp_path)
    base = dict(os.environ, PYTHONPATH=str(ROOT), QVERSE_EXPERIMENTS_DIR=str(exps), MASTER_ADDR="127.0.0.1", QVERSE_DIST_BACKEND="gloo")
    base.pop("WORLD_SIZE", None)
    args = ["-m", "offline_tarteel_amd.benchmark.runner", "--experiment", "stub-exp", "--corpus", str(corpus), "--batch", "3"]
    one = subprocess.run([sys.executable, *args], capture_output=True, text=True, cwd=str(ROOT), timeout=300,
                         env=dict(base, QVERSE_RESULTS_DIR=str(tmp_path / "res1")))
    assert one.returncode == 0, one.stdout + one.stderr
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29553" if deal == "strided" else "29555", *args, "--deal", deal],
                         capture_output=True, text=True, cwd=str(ROOT), timeout=300, env=dict(base, QVERSE_RESULTS_DIR=str(tmp_path / "res2")))
    assert two.returncode == 0, two.stdout + two.stderr

    def load(d):
        files = [f for f in sorted(d.glob("*.json")) if f.name != "latest.json"]
        assert len(files) == 1, files
        return json.loads(files[0].read_text()), json.loads((d / "latest.json").read_text())

    r1, l1 = load(tmp_path / "res1")
    r2, l2 = load(tmp_path / "res2")
    assert r2[0]["world_size"] == 2 and r1[0]["total"] == r2[0]["total"] == 11      # the absent file is not a row
    assert _strip(r1) == _strip(r2)
    ids = [row["id"] for row in r2[0]["per_sample"]]
    assert ids == [f"s{i}" for i in range(11)]                                    # manifest order restored
    assert sum(1 for row in r2[0]["per_sample"] if row.get("error")) == 1          # the undecodable file, and only it
    assert [row for row in r2[0]["per_sample"] if row["id"] == "s4"][0]["predicted"] == []
    assert any(len(row["predicted"]) > 1 for row in r2[0]["per_sample"])           # spans expand to one emission per ayah
    assert l1[0]["total"] == l2[0]["total"] and l1[0]["recall"] == l2[0]["recall"]


def test_shard_plan_deals():
    from offline_tarteel_amd.dist import shard_plan

    lens = [5, 30, 12, 7, 22, 9, 18]
    for world in (1, 2, 3, 8):
        for deal in ("strided", "contiguous"):
            order, slices = shard_plan(lens, world, deal)
            assert len(order) % world == 0 and sorted(int(i) for i in order if i >= 0) == list(range(len(lens)))
            shares = [[int(i) for i in order[s] if i >= 0] for s in slices]
            for sh in shares:       # every share is sorted longest first
                assert [lens[i] for i in sh] == sorted((lens[i] for i in sh), reverse=True)
            if deal == "strided" and world == 2:
                # balanced: the two shares' total lengths differ by less than the longest clip
                tot = [sum(lens[i] for i in sh) for sh in shares]
                assert abs(tot[0] - tot[1]) < max(lens)
    order, slices = shard_plan([], 4)
    assert len(order) == 0


def test_four_decimal_scores_survive_the_float32_rows():
    from offline_tarteel_amd.dist import pack_results, unpack_results

    vals = [k / 10000 for k in range(0, 10001)]
    rows = pack_results([{"surah": 1, "ayah": 1, "ayah_end": None, "score": v} for v in vals])
    back = unpack_results(rows, round_dp=4)
    assert [b["score"] for b in back] == vals


def test_sharded_runner_with_fewer_files_than_ranks_and_with_none(tmp_path):
    """edge cases of the sharded entry on two gloo ranks: ONE present file (rank 1's share is all padding) and NO present file
    (no collective is attempted: both ranks leave through the same door)."""
    corpus, exps = _stub_corpus(tmp_path)
    m = json.loads((corpus / "manifest.json").read_text())
    base = dict(os.environ, PYTHONPATH=str(ROOT), QVERSE_EXPERIMENTS_DIR=str(exps), MASTER_ADDR="127.0.0.1", QVERSE_DIST_BACKEND="gloo")
    for tag, keep, port in (("one", [m["samples"][0]], "29557"), ("none", [m["samples"][-1]], "29559")):     # the last row's file is absent
        (corpus / "manifest.json").write_text(json.dumps({"samples": keep}))
        out = tmp_path / f"res_{tag}"
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                            "--master-port", port, "-m", "offline_tarteel_amd.benchmark.runner", "--experiment", "stub-exp",
                            "--corpus", str(corpus), "--batch", "3"],
                           capture_output=True, text=True, cwd=str(ROOT), timeout=300, env=dict(base, QVERSE_RESULTS_DIR=str(out)))
        assert r.returncode == 0, r.stdout + r.stderr
        files = [f for f in sorted(out.glob("*.json")) if f.name != "latest.json"]
        doc = json.loads(files[0].read_text())
        assert doc[0]["total"] == (1 if tag == "one" else 0) and doc[0]["world_size"] == 2
        if tag == "one":
            assert doc[0]["per_sample"][0]["id"] == "s0"
