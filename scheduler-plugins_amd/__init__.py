"""scheduler-plugins_amd — MI355X-native batched Filter/Score engine for the kube-scheduler plugins'
hot path.  The product is libspx.so (HIP kernels behind the C ABI of include/spx.h);
this package is the thin Python host side used by tests and bench.py: it loads the library through
ctypes, exactly the way a cgo shim binds it, and never computes a score itself.

The directory name carries a hyphen (it mirrors the upstream project name); import it as
`scheduler_plugins_amd` — the repo-root module of that name loads this directory as the package.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

from ._abi import Header, Table

PKG_DIR = Path(__file__).resolve().parent
ROOT = PKG_DIR.parent
HEADER_PATH = ROOT / "include" / "spx.h"
LIB_PATH = PKG_DIR / "libspx.so"

_hdr: Optional[Header] = None
_lib: Optional[C.CDLL] = None


class SpxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"spx error {code}: {msg}")
        self.code = code
        self.msg = msg


def header() -> Header:
    global _hdr
    if _hdr is None:
        _hdr = Header(str(HEADER_PATH))
    return _hdr


def lib() -> C.CDLL:
    """Loads libspx.so; raises (never falls back) when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(f"{LIB_PATH} is missing: run `python __graft_entry__.py build` (hipcc, gfx950). "
                              "There is no CPU fallback for the product path.")
        _lib = C.CDLL(str(LIB_PATH), mode=os.RTLD_GLOBAL if hasattr(os, "RTLD_GLOBAL") else C.DEFAULT_MODE)
        missing = header().bind(_lib)
        if missing:
            raise ImportError(f"libspx.so does not export: {missing}")
    return _lib


from .engine import Engine, PLUGINS  # noqa: E402

__all__ = ["Engine", "PLUGINS", "Header", "Table", "SpxError", "header", "lib", "ROOT", "PKG_DIR"]
