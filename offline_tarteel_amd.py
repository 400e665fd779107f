"""Import shim: ``import offline_tarteel_amd`` loads the package that lives in the
(hyphenated, hence not directly importable) directory ``offline-tarteel_amd/`` --
the same trick the reference uses for its hyphenated experiment dirs
(benchmark/runner.py:89-94)."""

import importlib.util as _ilu
import sys as _sys
from pathlib import Path as _Path

_dir = _Path(__file__).resolve().parent / "offline-tarteel_amd"
_spec = _ilu.spec_from_file_location(
    "offline_tarteel_amd", str(_dir / "__init__.py"), submodule_search_locations=[str(_dir)]
)
_mod = _ilu.module_from_spec(_spec)
_sys.modules["offline_tarteel_amd"] = _mod
_spec.loader.exec_module(_mod)
