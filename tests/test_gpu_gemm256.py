"""The 256 x 256-tile GEMM (csrc/qv_gemm256.hip) against the 128-wide one (csrc/qv_gemm.hip).

Both kernels form the same f16 products and accumulate them in the same order (same MFMA
instruction, same operand order, K walked in the same 16-deep steps), so the tile shape must not
change a single bit of anything downstream: the log-probs of a whole forward are compared bit for
bit under the three tile policies (0 = 128-wide only, 1 = default, 2 = 256 x 256 wherever
N % 256 == 0 -- FFN-up/down, QKV incl. the transposed V store, out-projection, both pointwise
convolutions incl. GLU, the subsampling projection), on ragged batches whose row counts are not
multiples of 256.  The oracle comparison of the forward itself lives in test_gpu_forward.py."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[0, 1, 2], ids=["fp16", "int4_int8", "ort_mixed"])
def eng(request):
    """precision 0 = f16 weights, 1 = QV_PREC_MIXED_INT4_INT8 (block-128 int4 Linear layers + per-channel int8
    pointwise convolutions, dequantised inside the MFMA operand fetch of either tile shape), 2 = QV_PREC_ORT_MIXED
    (int4 Linear layers + every Conv as uint8 x int8 -> int32 on the i8 MFMA of either tile shape, same epilogue
    arithmetic and the same per-utterance activation ranges)."""
    import offline_tarteel_amd  # noqa: F401
    from offline_tarteel_amd.engine import Engine

    e = Engine(device=0, with_model=True, seed=11, max_batch=24, max_samples=160000, precision=request.param)
    e.mixed = bool(request.param)
    yield e
    e.gemm_tiles(-1)
    e.gemm_tile_height(-1)
    del e


def _forward_bits(eng, audio, lens, mode, height=0):
    import torch

    eng.gemm_tiles(mode)
    eng.gemm_tile_height(height)     # 0 = 256-row tiles, 3 = 192-row tiles wherever the wide kernel runs (round 6)
    lp, T = eng.forward(audio, lens)
    torch.cuda.synchronize()
    return lp.cpu().numpy().view(np.uint32).copy(), list(T)


@pytest.mark.parametrize("batch,samples", [(3, 48000), (24, 160000)])
def test_tile_shape_does_not_change_a_bit_of_the_logprobs(eng, batch, samples):
    import torch

    from synth import synth_audio

    audio = torch.from_numpy(synth_audio(batch, samples, seed=batch)).cuda()
    lens = [samples - 1600 * (b % 5) for b in range(batch)]   # ragged: packed row counts are not tile multiples
    for b in range(batch):
        audio[b, lens[b]:] = 0
    ref, T0 = _forward_bits(eng, audio, lens, 0)
    assert np.isfinite(ref.view(np.float32)).all()
    for mode, height in ((1, 0), (2, 0), (2, 3), (1, 2), (1, -1)):
        got, T = _forward_bits(eng, audio, lens, mode, height)
        assert T == T0
        assert np.array_equal(got, ref), f"tile policy {mode}, height {height}: {np.count_nonzero(got != ref)} words differ"


def test_replay_reports_the_kernel_the_policy_picks(eng):
    import torch

    from synth import synth_audio

    audio = torch.from_numpy(synth_audio(24, 160000, seed=2)).cuda()
    eng.gemm_tiles(0)
    eng.forward(audio, [160000] * 24)
    torch.cuda.synchronize()
    # (int4 weights with fewer than 400 tiles of 128: 64-wide tiles, see launch_gemm's plan)
    small = "64" if eng.mixed else "128"
    assert eng.replay_gemm(0, iters=2)["kernel"] == f"k_gemm<f16_swish,{small}>"
    assert eng.replay_gemm(1, iters=2)["kernel"] == f"k_gemm<resid,{small}>"
    eng.gemm_tiles(2)
    eng.gemm_tile_height(0)
    assert eng.replay_gemm(0, iters=2)["kernel"] == "k_gemm256<f16_swish>"
    assert eng.replay_gemm(1, iters=2)["kernel"] == "k_gemm256<resid>"
    eng.gemm_tile_height(3)
    assert eng.replay_gemm(0, iters=2)["kernel"] == "k_gemm256<f16_swish,192>"
    assert eng.replay_gemm(1, iters=2)["kernel"] == "k_gemm256<resid,192>"
    eng.gemm_tile_height(0)
    eng.gemm_tiles(1)   # 24 x 126 rows: 12 x 8 = 96 tiles of 256 x 256 for FFN-up -> below the 160-tile threshold
    assert eng.replay_gemm(0, iters=2)["kernel"] == f"k_gemm<f16_swish,{small}>"
    classes = {}
    eng.gemm_tiles(2)
    eng.profile_gemm(True)
    eng.forward(audio, [160000] * 24)
    torch.cuda.synchronize()
    for c in eng.profile_gemm_read():
        classes[c["kernel"]] = c["launches"]
    eng.profile_gemm(False)
    assert classes.get("k_gemm256<f16_swish>") == 34 and classes.get("k_gemm256<qkv>") == 17, classes
