"""qverse: MI355X-native offline Quran-verse recognition hot path.

Host-side mirror of the reference's c2c-direct-mixed plugin surface over a C-ABI HIP
library (csrc/ -> libqverse.so).  See DESIGN.md.
"""

import os
from pathlib import Path

# Engines with several batches in flight use one HIP stream per batch; the runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise (measured: four
# context streams 14.8 k utt/s on 4 queues, 17.3 k on 8).  Only effective if set before HIP initialises,
# i.e. import this package before the first torch.cuda call; an existing setting is left alone.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# Kernel arguments in device memory instead of host memory the command processor reads over PCIe: the path is ~250 short
# launches per batch (one batch at a time 12.05 k -> 12.7 k utt/s, four in flight +0.9 %).  Same rule: before HIP initialises.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

PKG_DIR = Path(__file__).resolve().parent
DATA_DIR = PKG_DIR / "data"
TABLES_PATH = DATA_DIR / "qverse_tables.bin"
LIB_PATH = PKG_DIR / "libqverse.so"

__all__ = ["PKG_DIR", "DATA_DIR", "TABLES_PATH", "LIB_PATH"]
