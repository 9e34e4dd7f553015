#!/usr/bin/env python3
"""round 3 experiments: builds tools/r3/_var/libspx_<name>.so = the in-tree objects with ONE translation unit recompiled under
extra -D flags.  usage: variant.py <name> <source under csrc/> [-DFOO=1 ...]"""
import subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
PKG = ROOT / "scheduler-plugins_amd"
name, src, *flags = sys.argv[1:]
out = ROOT / "tools/r3/_var"
out.mkdir(exist_ok=True)
obj = out / f"{name}.o"
common = ["-O3", "-std=c++17", "-fPIC", "-pthread", "-ffp-contract=off", "-fno-fast-math", f"-I{ROOT / 'include'}", "--offload-arch=gfx950"]
subprocess.check_call(["/opt/rocm/bin/hipcc", *common, *flags, "-c", str(PKG / "csrc" / src), "-o", str(obj)])
objs = [str(o) for o in sorted((PKG / "_obj").glob("*.o")) if o.name != src + ".o"] + [str(obj)]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-pthread", "--offload-arch=gfx950", "-o", str(out / f"libspx_{name}.so"), *objs])
print(out / f"libspx_{name}.so")
