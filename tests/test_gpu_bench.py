"""bench.py end to end on one GPU: the plain run and the torch.distributed.run launch with the
collective path forced on (one rank over RCCL), so the N > 1 plumbing -- process group on the
device, lagged all-gather of the packed rows joined on the collective's stream, max-over-ranks
timing -- is exercised before the driver runs it on 2/4/8 GPUs."""

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _line(out: str) -> dict:
    rows = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(rows) == 1, out[-2000:]
    return json.loads(rows[0])


def test_bench_single_process_contract():
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "4", "--warmup", "1", "--batch", "8",
                        "--seconds", "5", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["value"] > 0 and d["roofline"]["bound"] == "mfma"
    assert 0 < d["roofline"]["frac"] < 1
    # HBM traffic is quoted from a committed counter summary only for the shape it was taken at (M = 8064)
    assert d["roofline"]["traffic"] is None and "no committed counter summary" in d["roofline"]["traffic_source"]
    assert d["dtype"] == "f16" and "not a BASELINE.json configuration" in d["config"]["workload"]
    assert d["roofline"]["kernel"].startswith("k_gemm") and "batches in flight" in d["roofline"]["tile_policy"]
    # the verse-shaped replay legs: every transcript passes the text gate / every one fails it
    pl = d["post_logits"]
    assert pl["gate_pass"]["use_ctc_fraction"] == 0.0 and pl["gate_fail"]["use_ctc_fraction"] == 1.0
    assert pl["gate_fail"]["ms_per_batch"] > pl["gate_pass"]["ms_per_batch"] > 0


def test_bench_tta30_workload_small():
    """--workload tta30 (BASELINE configs[4]) at a reduced size: anchor pass, gate, GPU-resampled copies,
    decision rule; seeded random weights gate every clip."""
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "tta30", "--steps", "2", "--warmup", "1",
                        "--batch", "4", "--seconds", "6", "--no-cpu-baseline", "--no-post-logits"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert "TTA" in d["metric"] and d["config"]["tta_gated_fraction"] == 1.0 and d["value"] > 0
    assert "configs[4]" in d["config"]["workload"] and d["config"]["batches_in_flight"] == 3


def test_bench_under_torchrun_with_collective_path():
    env = dict(os.environ, QVERSE_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", str(ROOT / "bench.py"), "--gpus", "1",
                        "--steps", "5", "--warmup", "2", "--batch", "8", "--seconds", "5", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = _line(p.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["batches_in_flight"] == 4
