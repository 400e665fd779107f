"""qverse: MI355X-native offline Quran-verse recognition hot path.

Host-side mirror of the reference's c2c-direct-mixed plugin surface over a C-ABI HIP
library (csrc/ -> libqverse.so).  See DESIGN.md.
"""

from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
DATA_DIR = PKG_DIR / "data"
TABLES_PATH = DATA_DIR / "qverse_tables.bin"
LIB_PATH = PKG_DIR / "libqverse.so"

__all__ = ["PKG_DIR", "DATA_DIR", "TABLES_PATH", "LIB_PATH"]
