"""c2c-direct-mixed-tta on MI355X -- drop-in for experiments/c2c-direct-mixed-tta/run.py."""

import sys
from pathlib import Path

_ROOT = Path(__file__).resolve().parents[3]
if str(_ROOT) not in sys.path:
    sys.path.insert(0, str(_ROOT))

import offline_tarteel_amd  # noqa: E402,F401
from offline_tarteel_amd import plugin as _p  # noqa: E402

predict = _p.predict_tta
predict_batch = _p.predict_tta_batch   # used by `benchmark.runner --batch N`
transcribe = _p.transcribe
model_size = _p.model_size
