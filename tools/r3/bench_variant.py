#!/usr/bin/env python3
"""round 3 experiments: bench.py against tools/r3/_var/libspx_<SPX_VARIANT>.so"""
import os, sys, runpy
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import scheduler_plugins_amd as spx
if os.environ.get("SPX_VARIANT"):
    spx.LIB_PATH = ROOT / "tools/r3/_var" / f"libspx_{os.environ['SPX_VARIANT']}.so"
sys.argv = [str(ROOT / "bench.py")] + sys.argv[1:]
runpy.run_path(str(ROOT / "bench.py"), run_name="__main__")
