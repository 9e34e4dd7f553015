"""The C-ABI library loads and exports every symbol include/spx.h declares (CPU: no compute calls)."""
import ctypes as C
import re

import pytest

import scheduler_plugins_amd as spx


def test_library_exports_every_declared_symbol():
    lib = spx.lib()
    src = open(spx.HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    declared = set(re.findall(r"\b(spx_\w+)\s*\(", src))
    assert len(declared) >= 25
    parsed = set(spx.header().protos)
    assert declared == parsed, f"header parser missed {declared ^ parsed}"
    for name in sorted(declared):
        assert hasattr(lib, name), f"libspx.so does not export {name}"
    assert lib.spx_abi_version() == 1


def test_no_torch_types_in_signatures():
    # plain pointers and sizes only: every argument / return type is a ctypes scalar, a pointer to one, or a pointer to a
    # struct declared in spx.h (ctypes caches POINTER(T) classes under whichever module created them first, so the
    # check is on the type's nature, not on its __module__)
    import ctypes as C

    def plain(t, depth=0):
        if t is None or issubclass(t, C._SimpleCData):
            return True
        if issubclass(t, C._Pointer) and depth < 3:
            return plain(t._type_, depth + 1)
        return issubclass(t, C.Structure) and t.__name__.startswith("spx_")

    for name, (ret, args) in spx.header().protos.items():
        for a in [ret] + list(args):
            assert plain(a), (name, a)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no product source may reference oracle/."""
    import pathlib
    pkg = pathlib.Path(spx.PKG_DIR)
    bad = []
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.cc")) + list(pkg.rglob("*.h")):
        txt = f.read_text()
        if re.search(r"pyoracle|liboracle|spx_oracle\.h|orc_\w+\(", txt):
            bad.append(str(f))
    assert not bad, bad


def test_create_without_gpu_fails_loudly():
    try:
        e = spx.Engine(0)
    except spx.SpxError as err:
        assert err.code == -4 and "no CPU fallback" in err.msg
    else:  # on a GPU box this is simply a working engine
        e.close()


def test_host_entry_points_reject_null_arguments():
    """every host-side entry point (flatteners, comparator, eviction simulation, wire-format decoder) answers SPX_ERR_ARG to
    an all-NULL call instead of dereferencing — the cgo shim's mistakes must come back as error codes"""
    import ctypes as C

    import scheduler_plugins_amd as spx

    hdr, lib = spx.header(), spx.lib()
    prefixes = ("spx_flatten", "spx_toposort", "spx_nrt_post", "spx_ingest_nrt_json", "spx_ingest_quantity", "spx_ingest_create")
    names = [n for n in hdr.protos if n.startswith(prefixes)]
    assert len(names) >= 18
    for n in names:
        fn = getattr(lib, n)
        args = [0 if t in (C.c_int64, C.c_int32, C.c_int, C.c_uint32) else None for t in fn.argtypes]
        assert fn(*args) == hdr.consts["SPX_ERR_ARG"], n
