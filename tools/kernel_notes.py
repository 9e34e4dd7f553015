#!/usr/bin/env python3
"""Code-object notes of every kernel in the in-tree objects: VGPRs, SGPRs, spilled registers, scratch bytes, LDS, and the waves
per SIMD the VGPR count allows (gfx950: 512 VGPRs per SIMD lane, granule 8).   tools/kernel_notes.py [substring ...]"""
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OBJ = ROOT / "scheduler-plugins_amd" / "_obj"
LLVM = Path("/opt/rocm/lib/llvm/bin")


def code_object(obj: Path) -> bytes:
    """the gfx950 code object inside the offload bundle embedded in a hipcc host object (as build.device_code_hash walks it)"""
    b = obj.read_bytes()
    i = b.find(b"__CLANG_OFFLOAD_BUNDLE__")
    n, = struct.unpack_from("<Q", b, i + 24)
    p = i + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", b, p)
        p += 24
        triple = b[p:p + tl].decode()
        p += tl
        if "amdgcn" in triple:
            return b[i + off:i + off + size]
    return b""  # (a translation unit without kernels)


def notes(obj: Path):
    blob = code_object(obj)
    if not blob:
        return []
    with tempfile.TemporaryDirectory() as td:
        co = Path(td) / "co"
        co.write_bytes(blob)
        txt = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True).stdout
    out = []
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        out.append(dict(name=name, vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), vspill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count"),
                        scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size")))
    return out


if __name__ == "__main__":
    pats = sys.argv[1:]
    for obj in sorted(OBJ.glob("*.hip.o")):
        for k in notes(obj):
            if pats and not any(p in k["name"] for p in pats):
                continue
            v = int(k["vgpr"]) if k["vgpr"].isdigit() else 0
            waves = min(8, 512 // max(8, (v + 7) // 8 * 8)) if v else "?"
            short = re.sub(r"^void spx::(\(anonymous namespace\)::)?", "", k["name"])[:70]
            print(f"{obj.name[:-6]:28s} {short:70s} vgpr {k['vgpr']:>4} sgpr {k['sgpr']:>4} spill v{k['vspill']}/s{k['sspill']} scratch {k['scratch']:>5} lds {k['lds']:>6} waves {waves}")
