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
