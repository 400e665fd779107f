import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The reference-fixture parity files of the hot path run FIRST, so that under `-x` a failure in tooling-level tests
# (converter, bench harness) can never mask them (a red converter test left 24 parity tests unreached).
_GPU_ORDER = ["test_gpu_postlogits", "test_gpu_tracker", "test_gpu_tta", "test_gpu_audio", "test_gpu_forward",
              "test_gpu_fullsize", "test_gpu_gemm256", "test_gpu_ort_mixed", "test_gpu_bench"]


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        stem = Path(str(item.fspath)).stem
        return _GPU_ORDER.index(stem) if stem in _GPU_ORDER else len(_GPU_ORDER)

    items.sort(key=key)   # stable: order inside a file (and among the CPU files) is unchanged


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"
