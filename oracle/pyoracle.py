"""Python binding of the CPU oracle (oracle/_build/liboracle.so) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package never does.  It reuses the product's header parser purely as a ctypes convenience
(the oracle depends on the product's *interface definition*, include/spx.h, not on its code path).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import Optional

import numpy as np

ORACLE_DIR = Path(__file__).resolve().parent
ROOT = ORACLE_DIR.parent
LIB_PATH = ORACLE_DIR / "_build" / "liboracle.so"


import sys

if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
import scheduler_plugins_amd as _spx  # noqa: E402  (interface definitions + ctypes helpers only)

Header, Table = _spx.Header, _spx.Table

_hdr: Optional[Header] = None
_lib: Optional[C.CDLL] = None


def header() -> Header:
    global _hdr
    if _hdr is None:
        _hdr = _spx.header().derive(str(ORACLE_DIR / "spx_oracle.h"))
    return _hdr


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            subprocess.run(["make", "-C", str(ORACLE_DIR)], check=True, stdout=subprocess.DEVNULL)
        _lib = C.CDLL(str(LIB_PATH))
        names = [n for n in header().protos if n.startswith("orc_")]
        missing = header().bind(_lib, names)
        if missing:
            raise ImportError(f"liboracle.so does not export {missing}")
    return _lib


class Snapshot:
    """Bundle of object tables + plugin params handed to orc_score_rows."""

    def __init__(self, nodes: Table, pods: Table, rc: Optional[Table] = None, metrics: Optional[Table] = None,
                 assigned: Optional[Table] = None, alloc_params: Optional[Table] = None,
                 tlp_params: Optional[Table] = None, lvrb_params: Optional[Table] = None,
                 nrt: Optional[Table] = None, nrt_params: Optional[Table] = None,
                 appgroups: Optional[Table] = None, nettopo: Optional[Table] = None,
                 node_pods: Optional[Table] = None, lroc_params: Optional[Table] = None,
                 power_models: Optional[Table] = None):
        h = header()
        self.keep = dict(nodes=nodes, pods=pods, rc=rc, metrics=metrics, assigned=assigned, alloc_params=alloc_params,
                         tlp_params=tlp_params, lvrb_params=lvrb_params, nrt=nrt, nrt_params=nrt_params, appgroups=appgroups, nettopo=nettopo,
                         node_pods=node_pods, lroc_params=lroc_params, power_models=power_models)
        self.struct = h.structs["orc_snapshot"]()
        for k, v in self.keep.items():
            if v is not None:
                setattr(self.struct, k, C.pointer(v.struct))
        self.n_nodes = nodes.struct.n_nodes
        self.n_pods = pods.struct.n_pods

    def score_rows(self, plugin: int, row_begin: int = 0, row_end: Optional[int] = None,
                   mask: Optional[np.ndarray] = None, threads: int = 1, want_raw: bool = True, want_norm: bool = True):
        row_end = self.n_pods if row_end is None else row_end
        rows = row_end - row_begin
        raw = np.zeros((rows, self.n_nodes), dtype=np.int64) if want_raw else None
        norm = np.zeros((rows, self.n_nodes), dtype=np.int64) if want_norm else None
        i64p = C.POINTER(C.c_int64)
        mptr = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            mptr = mask.ctypes.data_as(C.POINTER(C.c_uint8))
        rc = lib().orc_score_rows(C.byref(self.struct), plugin, row_begin, row_end, mptr, threads,
                                  raw.ctypes.data_as(i64p) if want_raw else None,
                                  norm.ctypes.data_as(i64p) if want_norm else None)
        if rc != 0:
            raise RuntimeError(f"orc_score_rows failed: {rc}")
        return raw, norm

    def filter_rows(self, plugin: int, row_begin: int = 0, row_end: Optional[int] = None, threads: int = 1) -> np.ndarray:
        row_end = self.n_pods if row_end is None else row_end
        out = np.zeros((row_end - row_begin, self.n_nodes), dtype=np.uint8)
        rc = lib().orc_filter_rows(C.byref(self.struct), plugin, row_begin, row_end, threads,
                                   out.ctypes.data_as(C.POINTER(C.c_uint8)))
        if rc != 0:
            raise RuntimeError(f"orc_filter_rows failed: {rc}")
        return out


def usable_cpus() -> int:
    """CPUs this process can actually run on: sched_getaffinity capped by the cgroup's cpu.max (orc_usable_cpus)"""
    return int(lib().orc_usable_cpus())


def commit_sequential(snap: "Snapshot", plugin_mask: int, weights: Optional[dict] = None, quota: Optional[Table] = None,
                      row_begin: int = 0, row_end: Optional[int] = None, bind_ts: int = 0, threads: int = 0):
    """The one-pod-at-a-time cycle with its Reserve side effects (orc_commit_sequential): -> dict(node, score, ties, verdict,
    appended).  weights: {plugin id: weight} (missing = 1); threads 0 = every usable CPU."""
    h = header()
    row_end = snap.n_pods if row_end is None else row_end
    n = row_end - row_begin
    a = h.structs["orc_commit_args"]()
    a.s = C.pointer(snap.struct)
    if quota is not None:
        a.quota = C.pointer(quota.struct)
    a.plugin_mask = plugin_mask
    w = np.ones(h.consts["SPX_NUM_PLUGINS"], np.int64)
    for k, v in (weights or {}).items():
        w[k] = v
    a.weights = w.ctypes.data_as(C.POINTER(C.c_int64))
    a.row_begin, a.row_end, a.bind_ts = row_begin, row_end, bind_ts
    a.threads = threads if threads > 0 else usable_cpus()
    appended = np.zeros(snap.n_nodes, np.int32)
    a.tlp_appended_out = appended.ctypes.data_as(C.POINTER(C.c_int32))
    node, score, ties, verdict = np.zeros(n, np.int32), np.zeros(n, np.int64), np.zeros(n, np.int32), np.zeros(n, np.uint8)
    rc = lib().orc_commit_sequential(C.byref(a), node.ctypes.data_as(C.POINTER(C.c_int32)), score.ctypes.data_as(C.POINTER(C.c_int64)),
                                     ties.ctypes.data_as(C.POINTER(C.c_int32)), verdict.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc != 0:
        raise RuntimeError(f"orc_commit_sequential failed: {rc}")
    return dict(node=node, score=score, ties=ties, verdict=verdict, appended=appended)
