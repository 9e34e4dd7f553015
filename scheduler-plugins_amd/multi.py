"""ctypes wrapper over the spx_multi_* C ABI: one host process, several MI355X, pod rows sharded in equal contiguous
ranges, node tables replicated, RCCL all-gather of the decisions / of a global table afterwards (include/spx.h).

Every rank's engine is loaded by its own host thread: the node tables (replicated) and its slice of the pending batch — a view
of the pod object table — through the loaders a single engine uses; only CapacityScheduling's tables, whose nominated-pod
self-exclusion is by batch row, are flattened once and sliced."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from ._abi import Table
from .engine import Engine
from .objects import pod_rows

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
        self.load_ms = [0.0] * self.size  # per rank: host time (flatten + upload) of the load_* calls so far

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

    def rccl_ranks(self) -> int:
        """ranks of the RCCL communicator (ncclCommCount); 0 with the peer-copy transport"""
        n = int(self._lib.spx_multi_rccl_ranks(self._h))
        if n < 0:
            self._ck(n)
        return n

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

    # ------------------------------------------------------------------ tables: every rank flattens and uploads its own pod rows
    #
    # One host thread per rank (the library calls release the GIL): the rank's slice of the pending batch is a VIEW of the pod
    # object table — the per-pod columns offset, the per-container columns shared (their CSR offsets are absolute) — and goes
    # through the same one-call loaders a single engine uses, so the host work of a batch is spread over the ranks instead of
    # running once in front of them (config #5: 0.5 s of flatten for 500k pods before any device had work).  `load_ms[r]`: the
    # wall time rank r's thread has spent in load calls.  A rank whose shard is empty takes the whole-table path with an empty row range.
    def _per_rank(self, pods: Table, sharded, whole) -> None:
        """sharded(engine, pod_view) for ranks with rows, whole(engine, (b, b)) for ranks without; one thread per rank"""
        import threading
        import time
        n_total = pods.struct.n_pods
        work = list(self._each(n_total))
        errs: list = [None] * self.size

        def run(r, e, rows):
            t0 = time.perf_counter()
            try:
                if rows[1] > rows[0]:
                    sharded(e, pod_rows(self._hdr, pods, *rows))
                else:
                    whole(e, rows)
            except BaseException as ex:  # re-raised on the caller's thread
                errs[r] = ex
            self.load_ms[r] += (time.perf_counter() - t0) * 1e3

        threads = [threading.Thread(target=run, args=(r, e, rows)) for r, (e, rows) in enumerate(work)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for ex in errs:
            if ex is not None:
                raise ex

    def load_trimaran_objects(self, nodes: Table, rc, pods: Table, metrics: Table, assigned=None) -> None:
        def whole(e, rows):
            e.upload_alloc_nodes(e.flatten_alloc_nodes(nodes, rc))
            e.upload_trimaran_nodes(e.flatten_trimaran_nodes(nodes, metrics, assigned))
            e.upload_trimaran_pods(e.flatten_trimaran_pods(pods), rows)
        self._per_rank(pods, lambda e, view: e.load_trimaran_objects(nodes, rc, view, metrics, assigned), whole)
        self.n_nodes = nodes.struct.n_nodes

    def load_lroc_objects(self, nodes: Table, node_pods, pods: Table) -> None:
        def whole(e, rows):
            e.upload_lroc_nodes(e.flatten_lroc_nodes(nodes, node_pods))
            e.upload_lroc_pods(e.flatten_lroc_pods(pods), rows)
        self._per_rank(pods, lambda e, view: e.load_lroc_objects(nodes, node_pods, view), whole)

    def load_peaks_objects(self, nodes: Table, metrics: Table, power_models, pods: Table) -> None:
        self._per_rank(pods, lambda e, view: e.load_peaks_objects(nodes, metrics, power_models, view),
                       lambda e, rows: e.upload_peaks(e.flatten_peaks(nodes, metrics, power_models, pods), rows))
        self.n_nodes = nodes.struct.n_nodes

    def load_nrt_objects(self, nodes: Table, nrt: Table, rc, pods: Table, params: Table) -> None:
        # (the dense resource-slot numbering is built from the rank's own pods and the zones: it may differ between ranks, which
        # no result depends on — every rank's tables are self-contained)
        self._per_rank(pods, lambda e, view: e.load_nrt_objects(nodes, nrt, rc, view, params),
                       lambda e, rows: e.upload_nrt(e.flatten_nrt(nodes, nrt, rc, pods, params), rows))
        self.n_nodes = nodes.struct.n_nodes

    def load_network_objects(self, nodes: Table, pods: Table, appgroups: Table, nettopo: Table) -> None:
        self._per_rank(pods, lambda e, view: e.load_network_objects(nodes, view, appgroups, nettopo),
                       lambda e, rows: e.upload_network(e.flatten_network(nodes, pods, appgroups, nettopo), rows))
        self.n_nodes = nodes.struct.n_nodes
        # TopologicalSort is a sort of the whole queue: the ranks' topology orders, in row order
        self.net_topo_order = np.concatenate([e.net_soa["topo_order"][:e.n_pods] for e in self.engines]) if pods.struct.n_pods else np.zeros(0, np.int32)

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
