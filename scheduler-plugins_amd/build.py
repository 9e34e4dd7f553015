"""Builds libspx.so (HIP kernels + C-ABI engine + host flatteners) for gfx950, in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree to the GPU box.
-ffp-contract=off is load-bearing: the reference's float64 arithmetic (Go, amd64) never fuses
multiply-add, and the parity bar for Filter/Score is bit-exactness.
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
HOST = PKG / "host"
OBJ = PKG / "_obj"
LIB = PKG / "libspx.so"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
EXTRA = os.environ.get("SPX_EXTRA_CFLAGS", "").split()
COMMON = [*EXTRA, "-O3", "-std=c++17", "-fPIC", "-pthread", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
          f"-I{ROOT / 'include'}"]
DEVICE = ["--offload-arch=gfx950"]


def _sources():
    return sorted(CSRC.glob("*.hip")) + sorted(HOST.glob("*.cc"))


def _deps_mtime() -> float:
    hdrs = list(CSRC.glob("*.h")) + list(HOST.glob("*.h")) + list(HOST.glob("*.hpp")) + list((ROOT / "include").glob("*.h")) + [Path(__file__)]
    return max(h.stat().st_mtime for h in hdrs)


def build(verbose: bool = False, force: bool = False) -> Path:
    OBJ.mkdir(exist_ok=True)
    dep_m = _deps_mtime()
    objs, rebuilt = [], False
    procs = []
    for src in _sources():
        obj = OBJ / (src.name + ".o")
        objs.append(obj)
        if not force and obj.exists() and obj.stat().st_mtime >= max(src.stat().st_mtime, dep_m):
            continue
        cmd = [HIPCC, *COMMON]
        if src.suffix == ".hip":
            cmd += DEVICE
        else:
            cmd += ["-x", "c++"]
        cmd += ["-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        rebuilt = True
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{out}")
        if verbose and out.strip():
            print(out)
    if rebuilt or not LIB.exists() or force:
        cmd = [HIPCC, "-shared", "-fPIC", "-pthread", *DEVICE, "-o", str(LIB), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


def device_code_hash(sources, obj_dir: Path = OBJ) -> str:
    """sha256 over the gfx950 machine code (.text + .rodata of the code object embedded in <source>.o) of the given kernel
    translation units — what identifies "the kernels a profile was taken with": unlike a hash of the sources it does not move
    when a shared header gains a field some other translation unit uses.  Raises FileNotFoundError when an object is missing."""
    import hashlib
    import struct
    h = hashlib.sha256()
    for name in sorted(set(sources)):
        b = (Path(obj_dir) / (name + ".o")).read_bytes()
        i = b.find(b"__CLANG_OFFLOAD_BUNDLE__")
        if i < 0:
            raise ValueError(f"{name}.o holds no offload bundle")
        n, = struct.unpack_from("<Q", b, i + 24)
        p = i + 32
        found = False
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", b, p)
            p += 24
            triple = b[p:p + tl].decode()
            p += tl
            if "amdgcn" not in triple:
                continue
            co = b[i + off:i + off + size]
            shoff, = struct.unpack_from("<Q", co, 0x28)
            shentsize, shnum, shstrndx = struct.unpack_from("<HHH", co, 0x3A)
            secs = [struct.unpack_from("<IIQQQQIIQQ", co, shoff + j * shentsize) for j in range(shnum)]
            stroff = secs[shstrndx][4]
            for sec in secs:
                sname = co[stroff + sec[0]:co.index(b"\0", stroff + sec[0])].decode()
                if sname in (".text", ".rodata"):
                    h.update(name.encode() + sname.encode())
                    h.update(co[sec[4]:sec[4] + sec[5]])
            found = True
        if not found:
            raise ValueError(f"{name}.o holds no amdgcn code object")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
