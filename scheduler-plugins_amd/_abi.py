"""ctypes view of a C header kept in "one field per line" form (include/spx.h).

The headers are the single source of truth for the C ABI; this module parses their struct
typedefs and function prototypes and builds ctypes Structure classes / argtypes from them, so
the Python binding can never drift from what the Go shim would bind through cgo.
"""
from __future__ import annotations

import ctypes as C
import re
from typing import Dict, List, Tuple

import numpy as np

_SCALARS = {
    "int": C.c_int,
    "int8_t": C.c_int8,
    "int32_t": C.c_int32,
    "int64_t": C.c_int64,
    "uint8_t": C.c_uint8,
    "uint32_t": C.c_uint32,
    "uint64_t": C.c_uint64,
    "double": C.c_double,
    "float": C.c_float,
    "char": C.c_char,
    "void": None,
}

_NP = {
    C.c_int8: np.int8,
    C.c_int32: np.int32,
    C.c_int64: np.int64,
    C.c_uint8: np.uint8,
    C.c_uint32: np.uint32,
    C.c_double: np.float64,
    C.c_float: np.float32,
    C.c_int: np.int32,
}


def _strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return src


class Header:
    """Parsed header: .structs (name -> ctypes.Structure subclass), .consts, .protos."""

    def __init__(self, *paths: str):
        self.structs: Dict[str, type] = {}
        self.opaque: Dict[str, type] = {}
        self.consts: Dict[str, int] = {}
        self.protos: Dict[str, Tuple[object, List[object]]] = {}
        self.field_np: Dict[str, Dict[str, object]] = {}
        for p in paths:
            self._parse(open(p).read())

    def derive(self, *paths: str) -> "Header":
        """A new Header that shares this one's struct classes and parses further headers on top
        (so that two libraries sharing table structs see the same ctypes classes)."""
        h = Header()
        h.structs = dict(self.structs)
        h.opaque = dict(self.opaque)
        h.consts = dict(self.consts)
        h.protos = dict(self.protos)
        h.field_np = dict(self.field_np)
        for p in paths:
            h._parse(open(p).read(), skip_known=True)
        return h

    # -- type resolution -------------------------------------------------
    def _ctype(self, t: str):
        t = t.strip()
        t = re.sub(r"\bconst\b", "", t).strip()
        t = re.sub(r"\bstruct\b", "", t).strip()
        stars = t.count("*")
        base = t.replace("*", "").strip()
        if base in _SCALARS:
            ct = _SCALARS[base]
        elif base in self.structs:
            ct = self.structs[base]
        elif base in self.opaque:
            ct = self.opaque[base]
        else:
            raise KeyError(f"unknown C type {base!r}")
        if stars == 0:
            return ct
        if base == "char" and stars == 1:
            return C.c_char_p
        if ct is None:  # void*
            res = C.c_void_p
            stars -= 1
        else:
            res = C.POINTER(ct)
            stars -= 1
        for _ in range(stars):
            res = C.POINTER(res)
        return res

    def _parse(self, src: str, skip_known: bool = False) -> None:
        for m in re.finditer(r"#define\s+(\w+)\s+\(?(-?\d+)\)?\s*(?:/\*.*?\*/)?\s*$", src, flags=re.M):
            self.consts[m.group(1)] = int(m.group(2))
        src = _strip_comments(src)
        # opaque handles: typedef struct X X;
        for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", src):
            name = m.group(2)
            if name not in self.structs:
                self.opaque[name] = type(name, (C.Structure,), {})
        for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
            name = m.group(3)
            if skip_known and name in self.structs:
                continue
            fields = []
            npmap = {}
            for line in m.group(2).split(";"):
                line = line.strip()
                if not line:
                    continue
                fm = re.match(r"(.+?)(\w+)$", line)
                ftype, fname = fm.group(1), fm.group(2)
                ct = self._ctype(ftype)
                fields.append((fname, ct))
                if "*" in ftype:
                    base = re.sub(r"\bconst\b|\*", "", ftype).strip()
                    npmap[fname] = _NP.get(_SCALARS.get(base)) if base in _SCALARS else base
            cls = type(name, (C.Structure,), {"_fields_": fields})
            self.structs[name] = cls
            self.field_np[name] = npmap
            self.opaque.pop(name, None)
        # prototypes:  <ret> name(args);
        body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)
        body = re.sub(r"^\s*#[^\n]*$", "", body, flags=re.M)  # preprocessor lines are not part of a prototype
        for m in re.finditer(r"([\w\s\*]+?)\b(\w+)\s*\(([^()]*)\)\s*;", body):
            ret, fname, args = m.group(1).strip(), m.group(2), m.group(3).strip()
            if not ret or ret.startswith("typedef") or ret.startswith("#"):
                continue
            try:
                rct = self._ctype(ret)
                if args in ("", "void"):
                    argtypes = []
                else:
                    argtypes = []
                    for a in args.split(","):
                        a = a.strip()
                        am = re.match(r"(.+?)(\w+)$", a)
                        argtypes.append(self._ctype(am.group(1)))
            except KeyError:
                continue
            self.protos[fname] = (rct, argtypes)

    def bind(self, lib: C.CDLL, names=None) -> List[str]:
        """Set restype/argtypes on every prototype found in `lib`; returns the missing symbols."""
        missing = []
        for fname, (rct, argtypes) in self.protos.items():
            if names is not None and fname not in names:
                continue
            try:
                fn = getattr(lib, fname)
            except AttributeError:
                missing.append(fname)
                continue
            fn.restype = rct
            fn.argtypes = argtypes
        return missing


class Table:
    """A ctypes struct instance plus the numpy arrays that back its pointer fields.

    Keeps the arrays alive for as long as the struct is, and converts dtypes/contiguity so that
    what C sees is exactly what the header declares.
    """

    def __init__(self, header: Header, struct_name: str, **fields):
        cls = header.structs[struct_name]
        self.struct = cls()
        self.name = struct_name
        self._keep = {}
        npmap = header.field_np[struct_name]
        for fname, ct in cls._fields_:
            if fname not in fields or fields[fname] is None:
                continue
            v = fields[fname]
            if fname in npmap:
                want = npmap[fname]
                if isinstance(want, str):  # pointer to another struct
                    if isinstance(v, Table):
                        self._keep[fname] = v
                        setattr(self.struct, fname, C.pointer(v.struct))
                    else:
                        raise TypeError(f"{struct_name}.{fname} wants a Table of {want}")
                else:
                    arr = np.ascontiguousarray(v, dtype=want)
                    if arr.size == 0:
                        arr = np.zeros(1, dtype=want)  # never hand C a NULL for an empty column
                    self._keep[fname] = arr
                    setattr(self.struct, fname, arr.ctypes.data_as(ct))
            else:
                setattr(self.struct, fname, v)
        unknown = set(fields) - {f for f, _ in cls._fields_}
        if unknown:
            raise KeyError(f"{struct_name}: unknown fields {sorted(unknown)}")

    def ref(self):
        return C.byref(self.struct)

    def array(self, fname: str) -> np.ndarray:
        return self._keep[fname]
