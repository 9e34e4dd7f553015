#!/usr/bin/env python3
"""A/B experiments without switches in product code: builds tools/_var/libspx_<name>.so = the in-tree objects with ONE
translation unit replaced — recompiled under extra -D flags, or compiled from another source file (an older revision of the
kernel, a scratch copy).
  build:  variant.py build <name> <source under csrc/ that is replaced> [--from path/to/other.hip] [-DFOO=1 ...]
  run:    variant.py run <name> script.py [args...]      (the script sees that library as scheduler_plugins_amd's libspx.so;
                                                         name `base` = the in-tree library)"""
import runpy
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "scheduler-plugins_amd"
OUT = ROOT / "tools" / "_var"


def build(name, src, flags):
    OUT.mkdir(exist_ok=True)
    source = PKG / "csrc" / src
    if "--from" in flags:
        i = flags.index("--from")
        source = Path(flags[i + 1])
        flags = flags[:i] + flags[i + 2:]
    obj = OUT / f"{name}.o"
    common = ["-O3", "-std=c++17", "-fPIC", "-pthread", "-ffp-contract=off", "-fno-fast-math", f"-I{ROOT / 'include'}", f"-I{PKG / 'csrc'}",
              "--offload-arch=gfx950"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", *common, *flags, "-c", str(source), "-o", str(obj)])
    objs = [str(o) for o in sorted((PKG / "_obj").glob("*.o")) if o.name != src + ".o"] + [str(obj)]
    lib = OUT / f"libspx_{name}.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-pthread", "--offload-arch=gfx950", "-o", str(lib), *objs])
    print(lib)


def run(name, script, args):
    sys.path.insert(0, str(ROOT))
    import scheduler_plugins_amd as spx
    if name != "base":
        spx.LIB_PATH = OUT / f"libspx_{name}.so"
    sys.argv = [script, *args]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3], sys.argv[4:])
    elif sys.argv[1] == "run":
        run(sys.argv[2], sys.argv[3], sys.argv[4:])
    else:
        sys.exit(__doc__)
