"""c2c-direct-mixed on MI355X -- drop-in for the reference's experiments/c2c-direct-mixed/run.py
(same three callables, loaded by file path by benchmark/runner.py)."""

import sys
from pathlib import Path

_ROOT = Path(__file__).resolve().parents[3]
if str(_ROOT) not in sys.path:
    sys.path.insert(0, str(_ROOT))

import offline_tarteel_amd  # noqa: E402,F401
from offline_tarteel_amd import plugin as _p  # noqa: E402

predict = _p.predict
predict_batch = _p.predict_batch
transcribe = _p.transcribe
model_size = _p.model_size
