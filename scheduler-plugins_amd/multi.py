"""ctypes wrapper over the spx_multi_* C ABI: one host process, several MI355X, pod rows sharded in equal contiguous
ranges, node tables replicated, RCCL all-gather of the decisions / of a global table afterwards (include/spx.h).

The flatteners run once on the host (they do not touch a device); every rank's engine then receives the node columns and its
own slice of the pod columns."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from ._abi import Table
from .engine import Engine

RCCL, PEER_COPY = 0, 1  # SPX_MULTI_TRANSPORT_*


class MultiEngine:
    def __init__(self, devices: Sequence[int], transport: int = RCCL):
        from . import SpxError, header, lib

        self._lib = lib()
        self._hdr = header()
        self._err = SpxError
        self._h = C.POINTER(self._hdr.opaque["spx_multi"])()
        ids = (C.c_int * len(devices))(*devices)
        rc = self._lib.spx_multi_create(ids, len(devices), transport, C.byref(self._h))
        if rc != 0:
            msg = self._lib.spx_multi_last_error(None)
            raise SpxError(rc, msg.decode() if msg else "")
        self.size = len(devices)
        self.engines: List[Engine] = []
        for r in range(self.size):
            eh = C.POINTER(self._hdr.opaque["spx_engine"])()
            self._ck(self._lib.spx_multi_engine(self._h, r, C.byref(eh)))
            self.engines.append(Engine(_handle=eh))
        self.n_nodes = 0
        self.n_pods = 0  # of the whole batch

    def _ck(self, rc: int) -> None:
        if rc != 0:
            msg = self._lib.spx_multi_last_error(self._h)
            raise self._err(rc, msg.decode() if msg else "")

    def close(self) -> None:
        if self._h:
            for e in self.engines:
                e.close()  # handles are owned by the spx_multi
            self._lib.spx_multi_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ sharding
    def shard(self, rank: int, n_pods_total: Optional[int] = None):
        b, e = C.c_int64(), C.c_int64()
        self._ck(self._lib.spx_multi_shard(self._h, self.n_pods if n_pods_total is None else n_pods_total, rank, C.byref(b), C.byref(e)))
        return int(b.value), int(e.value)

    def _each(self, n_pods_total: int):
        self.n_pods = n_pods_total
        for r, e in enumerate(self.engines):
            yield e, self.shard(r, n_pods_total)

    def for_all(self, fn) -> None:
        """apply a parameter setter to every rank's engine, e.g. m.for_all(lambda e: e.set_tlp(50))"""
        for e in self.engines:
            fn(e)

    # ------------------------------------------------------------------ tables: flatten once, node columns to all, pod slices to each
    def load_trimaran_objects(self, nodes: Table, rc, pods: Table, metrics: Table, assigned=None) -> None:
        e0 = self.engines[0]
        alloc = e0.flatten_alloc_nodes(nodes, rc)
        ncols = e0.flatten_trimaran_nodes(nodes, metrics, assigned)
        pcols = e0.flatten_trimaran_pods(pods)
        for e, rows in self._each(pods.struct.n_pods):
            e.upload_alloc_nodes(alloc)
            e.upload_trimaran_nodes(ncols)
            e.upload_trimaran_pods(pcols, rows)
        self.n_nodes = nodes.struct.n_nodes

    def load_lroc_objects(self, nodes: Table, node_pods, pods: Table) -> None:
        e0 = self.engines[0]
        ncols, pcols = e0.flatten_lroc_nodes(nodes, node_pods), e0.flatten_lroc_pods(pods)
        for e, rows in self._each(pods.struct.n_pods):
            e.upload_lroc_nodes(ncols)
            e.upload_lroc_pods(pcols, rows)

    def load_peaks_objects(self, nodes: Table, metrics: Table, power_models, pods: Table) -> None:
        f = self.engines[0].flatten_peaks(nodes, metrics, power_models, pods)
        for e, rows in self._each(f["P"]):
            e.upload_peaks(f, rows)
        self.n_nodes = f["N"]

    def load_nrt_objects(self, nodes: Table, nrt: Table, rc, pods: Table, params: Table) -> None:
        f = self.engines[0].flatten_nrt(nodes, nrt, rc, pods, params)
        for e, rows in self._each(f["P"]):
            e.upload_nrt(f, rows)
        self.n_nodes = f["N"]

    def load_network_objects(self, nodes: Table, pods: Table, appgroups: Table, nettopo: Table) -> None:
        f = self.engines[0].flatten_network(nodes, pods, appgroups, nettopo)
        for e, rows in self._each(f["P"]):
            e.upload_network(f, rows)
        self.n_nodes = f["N"]
        self.net_topo_order = f["cols"]["topo_order"]  # of the whole batch (the queue sort is global)

    def sort_queue(self, pods: Table) -> np.ndarray:
        """TopologicalSort over the whole pending queue: a global sort of 16-byte keys, done on rank 0's device"""
        return self.engines[0].sort_queue(pods, topo_order=self.net_topo_order)

    def load_quota_objects(self, pods: Table, rc, quota: Table) -> None:
        f = self.engines[0].flatten_quota(pods, rc, quota)
        for e, rows in self._each(f["P"]):
            e.upload_quota(f, rows)

    # ------------------------------------------------------------------ evaluation (no collective)
    def eval(self, plugin_mask: int) -> None:
        self._ck(self._lib.spx_multi_eval(self._h, plugin_mask))

    def eval_best(self, plugin_mask: int) -> None:
        self._ck(self._lib.spx_multi_eval_best(self._h, plugin_mask))

    def decide(self, plugin_mask: int) -> None:
        self._ck(self._lib.spx_multi_decide(self._h, plugin_mask))

    def sync(self) -> None:
        self._ck(self._lib.spx_multi_sync(self._h))

    def mark(self, which: int) -> None:
        self._ck(self._lib.spx_multi_mark(self._h, which))

    def marked_ms(self):
        """(max over ranks, per-rank list) of the HIP-event time between mark(0) and mark(1)"""
        mx = C.c_float()
        per = (C.c_float * self.size)()
        self._ck(self._lib.spx_multi_marked_ms(self._h, C.byref(mx), per))
        return float(mx.value), [float(x) for x in per]

    def last_ms(self):
        """(eval, gather) HIP-event durations of the last launches, max over ranks"""
        a, b = C.c_float(), C.c_float()
        self._ck(self._lib.spx_multi_last_ms(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    # ------------------------------------------------------------------ exchange
    def gather_best(self):
        """all-gather of the per-pod decisions -> (node, weighted score, ties, feasible) indexed by batch row"""
        n = self.n_pods
        node, score = np.zeros(n, np.int32), np.zeros(n, np.int64)
        ties, feas = np.zeros(n, np.int32), np.zeros(n, np.int32)
        i32p = C.POINTER(C.c_int32)
        self._ck(self._lib.spx_multi_gather_best(self._h, n, node.ctypes.data_as(i32p), score.ctypes.data_as(C.POINTER(C.c_int64)),
                                                 ties.ctypes.data_as(i32p), feas.ctypes.data_as(i32p)))
        return node, score, ties, feas

    def bind_global_table(self, plugin: int, status: bool = False) -> None:
        """call after the tables are loaded and before eval: each rank then writes its rows straight into its slice"""
        self._ck(self._lib.spx_multi_bind_global_table(self._h, plugin, 1 if status else 0, self.n_pods))

    def allgather_table(self, plugin: int, status: bool = False) -> None:
        self._ck(self._lib.spx_multi_allgather_table(self._h, plugin, 1 if status else 0))

    def global_rows(self, plugin: int, rank: int = 0, row_begin: int = 0, row_end: Optional[int] = None, status: bool = False) -> np.ndarray:
        row_end = self.n_pods if row_end is None else row_end
        out = np.empty((row_end - row_begin, self.n_nodes), dtype=np.uint8)
        self._ck(self._lib.spx_multi_fetch_global_rows(self._h, plugin, 1 if status else 0, rank, row_begin, row_end,
                                                       out.ctypes.data_as(C.POINTER(C.c_uint8)), self.n_nodes))
        return out
