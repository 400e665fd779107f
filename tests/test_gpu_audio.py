"""a15 on the GPU: qv_upfirdn (the polyphase FIR behind the TTA wrapper's speed perturbation,
experiments/c2c-direct-mixed-tta/run.py:60-71) against scipy.signal.resample_poly -- the very
call the reference makes -- bit for bit, through the C ABI."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from offline_tarteel_amd.engine import Engine

    e = Engine(device=0, with_model=False, max_batch=2, max_samples=16000)
    yield e
    e.close()


@pytest.mark.parametrize("n_in,up,down", [(160000, 9, 10), (160000, 11, 10), (48001, 9, 10), (777, 11, 10), (5, 9, 10),
                                          (1, 11, 10), (44100, 160, 441), (480000, 11, 10), (1000, 18, 20)])
def test_upfirdn_equals_scipy_resample_poly(eng, n_in, up, down):
    from scipy.signal import resample_poly

    rng = np.random.default_rng(n_in + up)
    x = (rng.standard_normal(n_in) * 0.3).astype(np.float32)
    want = resample_poly(x, up, down)
    got = eng.resample_poly(torch.from_numpy(x).cuda(), up, down).cpu().numpy()
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_speed_perturb_factors(eng):
    from scipy.signal import resample_poly

    from synth import synth_audio

    x = synth_audio(1, 80000, seed=9)[0]
    d = torch.from_numpy(x).cuda()
    assert eng.speed_perturb(d, 1.0) is d
    for f in (0.9, 1.1):
        want = resample_poly(x, int(f * 10), 10).astype("float32")   # tta/run.py:69-71
        assert np.array_equal(eng.speed_perturb(d, f).cpu().numpy(), want)
    # identity ratio after gcd reduction
    assert torch.equal(eng.resample_poly(d, 10, 10), d)


@pytest.mark.parametrize("up,down", [(9, 10), (11, 10), (160, 441), (1, 3)])
def test_batched_resampler_equals_scipy_row_by_row(eng, up, down):
    """qv_upfirdn_batch (round 6): many ragged rows in ONE launch, optionally picked out of a larger matrix (src_rows), written
    zero-padded into the engine's input layout -- every row bit-identical to scipy.signal.resample_poly of that row."""
    from scipy.signal import resample_poly

    rng = np.random.default_rng(up * 1000 + down)
    lens = [44100, 30000, 1, 7, 12345, 44099, 2205]
    x = np.zeros((len(lens) + 2, max(lens) + 13), dtype=np.float32)       # pitch > longest row, two rows never used
    for r, n in enumerate(lens):
        x[r + 1, :n] = (rng.standard_normal(n) * 0.3).astype(np.float32)
    src = [r + 1 for r in range(len(lens))]
    order = [3, 0, 6, 2, 5, 1, 4]                                          # output rows in another order than the source rows
    y, n_out = eng.resample_rows(torch.from_numpy(x).cuda(), [lens[i] for i in order], up, down, src_rows=[src[i] for i in order])
    y = y.cpu().numpy()
    for k, i in enumerate(order):
        want = resample_poly(x[src[i], : lens[i]], up, down)
        assert n_out[k] == len(want)
        assert np.array_equal(y[k, : n_out[k]].view(np.uint32), want.view(np.uint32)), (up, down, lens[i])
        assert not y[k, n_out[k]:].any()                                   # zero padding up to the pitch
    # without src_rows: row r of the input
    y2, n2 = eng.resample_rows(torch.from_numpy(x[1:1 + len(lens)]).cuda(), lens, up, down, out_pitch=max(n_out) + 5)
    y2 = y2.cpu().numpy()
    for r, n in enumerate(lens):
        want = resample_poly(x[r + 1, :n], up, down)
        assert np.array_equal(y2[r, : n2[r]].view(np.uint32), want.view(np.uint32)) and not y2[r, n2[r]:].any()


@pytest.mark.parametrize("channels", [2, 3, 6])
def test_device_mixdown_equals_numpy_mean(eng, channels):
    rng = np.random.default_rng(channels)
    frames = [1000, 1, 777]
    x = np.zeros((3, max(frames) * channels), dtype=np.float32)
    for r, n in enumerate(frames):
        x[r, : n * channels] = (rng.standard_normal(n * channels) * 0.5).astype(np.float32)
    y = eng.mixdown_rows(torch.from_numpy(x).cuda(), frames, channels).cpu().numpy()
    for r, n in enumerate(frames):
        want = x[r, : n * channels].reshape(-1, channels).mean(axis=1)
        assert np.array_equal(y[r, :n].view(np.uint32), want.astype(np.float32).view(np.uint32))
        assert not y[r, n:].any()


def _wav(path, x, sr, ch, kind):
    import struct

    if kind == "pcm16":
        pcm = np.clip(np.round(x * 32767.0), -32768, 32767).astype("<i2").tobytes()
        tag, bits = 1, 16
    else:
        pcm = x.astype("<f4").tobytes()
        tag, bits = 3, 32
    align = ch * bits // 8
    hdr = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, tag, ch, sr, sr * align, align, bits)
    path.write_bytes(hdr + b"data" + struct.pack("<I", len(pcm)) + pcm)


def test_device_ingest_equals_host_load_audio(eng, tmp_path):
    """audio.load_audio_device: containers parsed on the host, mix-down + resampling to 16 kHz on the GPU (44.1 kHz stereo
    -> 160/441, 48 kHz -> 1/3, 22.05 kHz 3-channel, 16 kHz mono untouched) -- the same float32 samples load_audio() returns,
    bit for bit, zero-padded to the longest clip, in the order of the paths."""
    from offline_tarteel_amd.audio import load_audio, load_audio_device, probe_samples

    rng = np.random.default_rng(44)
    specs = [(44100, 2, "pcm16", 30011), (16000, 1, "pcm16", 9000), (48000, 1, "f32", 24001), (22050, 3, "f32", 5000),
             (44100, 2, "pcm16", 1234), (16000, 2, "f32", 4000)]
    paths = []
    for k, (sr, ch, kind, n) in enumerate(specs):
        p = tmp_path / f"c{k}.wav"
        _wav(p, (rng.standard_normal(n * ch) * 0.2).astype(np.float32), sr, ch, kind)
        paths.append(str(p))
    dev, lens = load_audio_device(paths, eng)
    dev = dev.cpu().numpy()
    assert dev.shape == (len(paths), max(lens))
    for k, p in enumerate(paths):
        want = load_audio(p)
        assert lens[k] == len(want), (k, lens[k], len(want))
        assert np.array_equal(dev[k, : lens[k]].view(np.uint32), want.view(np.uint32)), specs[k]
        assert not dev[k, lens[k]:].any()
        assert abs(probe_samples(p) - len(want)) <= 1                       # the header probe the sharded runner sorts by
