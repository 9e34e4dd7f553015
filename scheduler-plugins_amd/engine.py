"""ctypes wrapper over the spx_* C ABI (include/spx.h).  One Engine == one spx_engine == one
scheduler profile on one GPU.  Every method is a direct call into libspx.so; errors raise."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np

from ._abi import Table

PLUGINS = {
    "NodeResourcesAllocatable": 0,
    "TargetLoadPacking": 1,
    "LoadVariationRiskBalancing": 2,
    "NodeResourceTopologyMatch": 3,
    "NetworkOverhead": 4,
    "CapacityScheduling": 5,
    "TopologicalSort": 6,
}
ALLOCATABLE, TLP, LVRB, NRT, NETOVERHEAD, CAPACITY, TOPOSORT, LROC, PEAKS = range(9)
NUM_PLUGINS = 9  # SPX_NUM_PLUGINS


def mask_of(*plugins: int) -> int:
    m = 0
    for p in plugins:
        m |= 1 << p
    return m


def _rows(cols: Dict[str, np.ndarray], n_total: int, rows) -> Dict[str, np.ndarray]:
    """slice of per-pod SoA columns: every column holds a fixed number of entries per pod, pod-major"""
    if rows is None:
        return cols
    b, e = rows
    out = {}
    for k, v in cols.items():
        per = len(v) // max(n_total, 1)
        out[k] = np.ascontiguousarray(v[b * per:e * per]) if e > b else np.zeros(max(per, 1), v.dtype)
    return out


class Engine:
    def __init__(self, device: int = 0, _handle=None):
        from . import SpxError, header, lib

        self._lib = lib()
        self._hdr = header()
        self._err = SpxError
        self._owned = _handle is None
        if _handle is not None:  # an engine owned by a spx_multi (MultiEngine)
            self._h = _handle
        else:
            self._h = C.POINTER(self._hdr.opaque["spx_engine"])()
            rc = self._lib.spx_create(device, C.byref(self._h))
            if rc != 0:
                msg = self._lib.spx_last_error(None)
                raise SpxError(rc, msg.decode() if msg else "")
        self.n_nodes = 0
        self.n_pods = 0
        self.alloc_params: Optional[Table] = None
        self.tlp_params = Table(self._hdr, "spx_tlp_params", target_utilization=40, default_requests_milli=1000,
                                requests_multiplier=1.5)
        self.lvrb_params = Table(self._hdr, "spx_lvrb_params", safe_variance_margin=1.0, safe_variance_sensitivity=1.0)
        self.set_allocatable()

    # ------------------------------------------------------------------ plumbing
    def _ck(self, rc: int) -> None:
        if rc != 0:
            msg = self._lib.spx_last_error(self._h)
            raise self._err(rc, msg.decode() if msg else "")

    def close(self) -> None:
        if self._h:
            if self._owned:
                self._lib.spx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ------------------------------------------------------------------ options (per engine, spx_set_option)
    def set_option(self, option, value: int) -> None:
        """option: an SPX_OPT_* id or its name without the prefix, e.g. "REFERENCE_KERNELS"."""
        if isinstance(option, str):
            option = self._hdr.consts["SPX_OPT_" + option]
        self._ck(self._lib.spx_set_option(self._h, option, int(value)))

    def get_option(self, option) -> int:
        if isinstance(option, str):
            option = self._hdr.consts["SPX_OPT_" + option]
        v = C.c_int64()
        self._ck(self._lib.spx_get_option(self._h, option, C.byref(v)))
        return int(v.value)

    def nrt_pod_classes(self):
        """(representative rows, copied rows) of the uploaded NRT pod batch (spx_nrt_pod_classes); (n_pods, 0) when no two pods agree"""
        u, d = C.c_int64(), C.c_int64()
        self._ck(self._lib.spx_nrt_pod_classes(self._h, C.byref(u), C.byref(d)))
        return int(u.value), int(d.value)

    def peaks_pod_classes(self):
        """(rows that are the first with their cpu request, rows that repeat one) of the uploaded Peaks pod batch (spx_peaks_pod_classes)"""
        u, d = C.c_int64(), C.c_int64()
        self._ck(self._lib.spx_peaks_pod_classes(self._h, C.byref(u), C.byref(d)))
        return int(u.value), int(d.value)

    def force_reference_kernels(self, *plugins: int) -> None:
        """run the reference-arithmetic sweep for these plugins (differential tests); no argument = back to the fast forms"""
        self.set_option("REFERENCE_KERNELS", mask_of(*plugins))

    def stats(self, reset: bool = False) -> np.ndarray:
        """cells re-evaluated by the exact float64 fallback of the fast sweeps, per plugin id (spx_fetch_stats)"""
        out = np.zeros(NUM_PLUGINS, np.int64)
        self._ck(self._lib.spx_fetch_stats(self._h, out.ctypes.data_as(C.POINTER(C.c_int64)), 1 if reset else 0))
        return out

    # ------------------------------------------------------------------ params
    def set_allocatable(self, mode: str = "Least", resources: Optional[Dict[int, int]] = None) -> None:
        """resources: {resource id: weight}; default = {memory: 1, cpu: 1<<20} (resource_allocation.go:36)."""
        if resources is None:
            resources = {1: 1, 0: 1 << 20}
        self.alloc_params = Table(self._hdr, "spx_allocatable_params", mode={"Least": 0, "Most": 1}[mode],
                                  n_res=len(resources), res=np.array(list(resources.keys()), dtype=np.int32),
                                  weight=np.array(list(resources.values()), dtype=np.int64))
        self._ck(self._lib.spx_set_allocatable_params(self._h, self.alloc_params.ref()))

    def set_tlp(self, target_utilization: int = 40, default_requests_milli: int = 1000, requests_multiplier: float = 1.5):
        self.tlp_params = Table(self._hdr, "spx_tlp_params", target_utilization=target_utilization,
                                default_requests_milli=default_requests_milli, requests_multiplier=requests_multiplier)
        self._ck(self._lib.spx_set_tlp_params(self._h, self.tlp_params.ref()))

    def set_lvrb(self, margin: float = 1.0, sensitivity: float = 1.0):
        self.lvrb_params = Table(self._hdr, "spx_lvrb_params", safe_variance_margin=margin,
                                 safe_variance_sensitivity=sensitivity)
        self._ck(self._lib.spx_set_lvrb_params(self._h, self.lvrb_params.ref()))

    def set_lroc(self, smoothing_window_size: int = 5, w_cpu: float = 0.5, w_mem: float = 0.5):
        self.lroc_params = Table(self._hdr, "spx_lroc_params", smoothing_window_size=smoothing_window_size,
                                 risk_limit_weight_cpu=w_cpu, risk_limit_weight_mem=w_mem)
        self._ck(self._lib.spx_set_lroc_params(self._h, self.lroc_params.ref()))

    # ------------------------------------------------------------------ flatten (host C++) + upload
    def flatten_alloc_nodes(self, nodes: Table, rc: Optional[Table]) -> np.ndarray:
        n = nodes.struct.n_nodes
        r = self.alloc_params.struct.n_res
        out = np.zeros((r, n), dtype=np.int64)
        self._ck(self._lib.spx_flatten_alloc_nodes(nodes.ref(), rc.ref() if rc else None, self.alloc_params.ref(),
                                                    out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def flatten_trimaran_nodes(self, nodes: Table, metrics: Table, assigned: Optional[Table]) -> Dict[str, np.ndarray]:
        n = nodes.struct.n_nodes
        cols = {
            "cap_cpu_milli": np.zeros(n, np.int64), "tlp_cpu_util": np.zeros(n, np.float64),
            "tlp_missing_milli": np.zeros(n, np.int64), "tlp_valid": np.zeros(n, np.uint8),
            "lv_alloc_cpu_milli": np.zeros(n, np.int64), "lv_alloc_mem": np.zeros(n, np.int64),
            "lv_cpu_avg": np.zeros(n, np.float64), "lv_cpu_std": np.zeros(n, np.float64),
            "lv_mem_avg": np.zeros(n, np.float64), "lv_mem_std": np.zeros(n, np.float64),
            "lv_flags": np.zeros(n, np.uint8),
        }
        fn = self._lib.spx_flatten_trimaran_nodes
        ptrs = [v.ctypes.data_as(t) for v, t in zip(cols.values(), fn.argtypes[4:])]
        self._ck(fn(nodes.ref(), metrics.ref(), assigned.ref() if assigned else None, self.tlp_params.ref(), *ptrs))
        return cols

    def flatten_trimaran_node_rows(self, nodes: Table, metrics: Table, assigned: Optional[Table], idx) -> Dict[str, np.ndarray]:
        """flatten_trimaran_nodes() for the listed nodes only (row j = node idx[j]): the input of update_trimaran_node_rows"""
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        n = len(idx)
        cols = {
            "cap_cpu_milli": np.zeros(n, np.int64), "tlp_cpu_util": np.zeros(n, np.float64),
            "tlp_missing_milli": np.zeros(n, np.int64), "tlp_valid": np.zeros(n, np.uint8),
            "lv_alloc_cpu_milli": np.zeros(n, np.int64), "lv_alloc_mem": np.zeros(n, np.int64),
            "lv_cpu_avg": np.zeros(n, np.float64), "lv_cpu_std": np.zeros(n, np.float64),
            "lv_mem_avg": np.zeros(n, np.float64), "lv_mem_std": np.zeros(n, np.float64),
            "lv_flags": np.zeros(n, np.uint8),
        }
        fn = self._lib.spx_flatten_trimaran_node_rows
        ptrs = [v.ctypes.data_as(t) for v, t in zip(cols.values(), fn.argtypes[6:])]
        self._ck(fn(nodes.ref(), metrics.ref(), assigned.ref() if assigned else None, self.tlp_params.ref(),
                    idx.ctypes.data_as(C.POINTER(C.c_int64)), n, *ptrs))
        return cols

    def flatten_trimaran_pods(self, pods: Table) -> Dict[str, np.ndarray]:
        p = pods.struct.n_pods
        cols = {"tlp_pod_milli": np.zeros(p, np.int64), "lv_req_cpu_milli": np.zeros(p, np.int64),
                "lv_req_mem": np.zeros(p, np.int64)}
        i64p = C.POINTER(C.c_int64)
        self._ck(self._lib.spx_flatten_trimaran_pods(pods.ref(), self.tlp_params.ref(),
                                                      *[v.ctypes.data_as(i64p) for v in cols.values()]))
        return cols

    def upload_alloc_nodes(self, alloc: np.ndarray) -> None:
        alloc = np.ascontiguousarray(alloc, dtype=np.int64)
        t = Table(self._hdr, "spx_alloc_nodes_soa", n_nodes=alloc.shape[1], n_res=alloc.shape[0], alloc=alloc)
        self._ck(self._lib.spx_upload_alloc_nodes(self._h, t.ref()))
        self.n_nodes = alloc.shape[1]

    def upload_trimaran_nodes(self, cols: Dict[str, np.ndarray]) -> None:
        n = len(cols["cap_cpu_milli"])
        t = Table(self._hdr, "spx_trimaran_nodes_soa", n_nodes=n, **cols)
        self._ck(self._lib.spx_upload_trimaran_nodes(self._h, t.ref()))
        self.n_nodes = n

    def update_trimaran_nodes(self, idx, cols: Dict[str, np.ndarray]) -> None:
        """rows `idx` of the trimaran node table replaced in place: cols = flatten_trimaran_nodes()'s columns for ALL nodes, of
        which only rows idx travel (spx_update_trimaran_nodes)"""
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        sub = {k: np.ascontiguousarray(v[idx]) for k, v in cols.items()}
        self._ck(self._lib.spx_update_trimaran_nodes(self._h, idx.ctypes.data_as(C.POINTER(C.c_int64)),
                                                     Table(self._hdr, "spx_trimaran_nodes_soa", n_nodes=len(idx), **sub).ref()))

    def update_trimaran_node_rows(self, idx, rows: Dict[str, np.ndarray]) -> None:
        """rows `idx` of the trimaran node table replaced in place: rows = flatten_trimaran_node_rows()'s columns (len(idx) rows)"""
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        self._ck(self._lib.spx_update_trimaran_nodes(self._h, idx.ctypes.data_as(C.POINTER(C.c_int64)),
                                                     Table(self._hdr, "spx_trimaran_nodes_soa", n_nodes=len(idx), **rows).ref()))

    def update_nrt_nodes(self, idx, f: dict) -> None:
        """rows `idx` of the NRT node tables replaced in place: f = flatten_nrt()'s result for the NEW snapshot
        (spx_update_nrt_nodes; the derived float64 columns are recomputed on the device for those nodes)"""
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        N, R = f["N"], f["R"]
        per = {"flags": 1, "max_numa": 1, "n_zones": 1, "zone_id": 8, "zone_present": 8, "zone_avail": 8 * max(R, 1), "zone_cost": 64,
               "min_avg_dist": 8, "node_present": 1}
        sub = {k: np.ascontiguousarray(f["nodes"][k].reshape(N, per[k])[idx].reshape(-1)) for k in per}
        self._ck(self._lib.spx_update_nrt_nodes(self._h, idx.ctypes.data_as(C.POINTER(C.c_int64)),
                                                Table(self._hdr, "spx_nrt_nodes_soa", n_nodes=len(idx), n_res=R, **sub).ref()))

    def upload_trimaran_pods(self, cols: Dict[str, np.ndarray], rows=None) -> None:
        cols = _rows(cols, len(cols["tlp_pod_milli"]), rows)
        p = len(cols["tlp_pod_milli"]) if rows is None else rows[1] - rows[0]
        if p > 0:
            t = Table(self._hdr, "spx_trimaran_pods_soa", n_pods=p, **cols)
            self._ck(self._lib.spx_upload_trimaran_pods(self._h, t.ref()))
        self.n_pods = p

    def load_trimaran_objects(self, nodes: Table, rc: Optional[Table], pods: Table, metrics: Table,
                              assigned: Optional[Table] = None) -> None:
        """objects -> (host flatten) -> SoA -> HBM, for Allocatable + TLP + LVRB."""
        self.upload_alloc_nodes(self.flatten_alloc_nodes(nodes, rc))
        self.upload_trimaran_nodes(self.flatten_trimaran_nodes(nodes, metrics, assigned))
        self.upload_trimaran_pods(self.flatten_trimaran_pods(pods))

    # ------------------------------------------------------------------ LowRiskOverCommitment
    _LROC_COLS = ("req_cpu_milli", "req_mem", "lim_cpu_milli", "lim_mem")

    def flatten_lroc_nodes(self, nodes: Table, node_pods: Optional[Table]) -> Dict[str, np.ndarray]:
        cols = {k: np.zeros(nodes.struct.n_nodes, np.int64) for k in self._LROC_COLS}
        i64p = C.POINTER(C.c_int64)
        self._ck(self._lib.spx_flatten_lroc_nodes(nodes.ref(), node_pods.ref() if node_pods else None,
                                                   *[v.ctypes.data_as(i64p) for v in cols.values()]))
        return cols

    def flatten_lroc_pods(self, pods: Table) -> Dict[str, np.ndarray]:
        cols = {k: np.zeros(pods.struct.n_pods, np.int64) for k in self._LROC_COLS}
        i64p = C.POINTER(C.c_int64)
        self._ck(self._lib.spx_flatten_lroc_pods(pods.ref(), *[v.ctypes.data_as(i64p) for v in cols.values()]))
        return cols

    def upload_lroc_nodes(self, cols: Dict[str, np.ndarray]) -> None:
        t = Table(self._hdr, "spx_lroc_nodes_soa", n_nodes=len(cols["req_mem"]), **cols)
        self._ck(self._lib.spx_upload_lroc_nodes(self._h, t.ref()))

    def upload_lroc_pods(self, cols: Dict[str, np.ndarray], rows=None) -> None:
        cols = _rows(cols, len(cols["req_mem"]), rows)
        p = len(cols["req_mem"]) if rows is None else rows[1] - rows[0]
        if p > 0:
            t = Table(self._hdr, "spx_lroc_pods_soa", n_pods=p, **cols)
            self._ck(self._lib.spx_upload_lroc_pods(self._h, t.ref()))
        self.n_pods = p

    def load_lroc_objects(self, nodes: Table, node_pods: Optional[Table], pods: Table) -> None:
        """LowRiskOverCommitment's own tables; the trimaran node table (metrics, allocatable) must be loaded already."""
        self.upload_lroc_nodes(self.flatten_lroc_nodes(nodes, node_pods))
        self.upload_lroc_pods(self.flatten_lroc_pods(pods))

    # ------------------------------------------------------------------ Peaks
    def flatten_peaks(self, nodes: Table, metrics: Table, power_models: Optional[Table], pods: Table) -> dict:
        n, p = nodes.struct.n_nodes, pods.struct.n_pods
        cols = {"cap_cpu_milli": np.zeros(n, np.int64), "cpu_util": np.zeros(n, np.float64), "valid": np.zeros(n, np.uint8),
                "k1": np.zeros(n, np.float64), "k2": np.zeros(n, np.float64)}
        fn = self._lib.spx_flatten_peaks_nodes
        self._ck(fn(nodes.ref(), metrics.ref(), power_models.ref() if power_models else None,
                    *[v.ctypes.data_as(t) for v, t in zip(cols.values(), fn.argtypes[3:])]))
        cpu = np.zeros(p, np.int64)
        self._ck(self._lib.spx_flatten_peaks_pods(pods.ref(), cpu.ctypes.data_as(C.POINTER(C.c_int64))))
        return {"nodes": cols, "pods": {"cpu_milli": cpu}, "N": n, "P": p}

    def upload_peaks(self, f: dict, rows=None) -> None:
        pc = _rows(f["pods"], f["P"], rows)
        p = f["P"] if rows is None else rows[1] - rows[0]
        self._ck(self._lib.spx_upload_peaks_nodes(self._h, Table(self._hdr, "spx_peaks_nodes_soa", n_nodes=f["N"], **f["nodes"]).ref()))
        if p > 0:
            self._ck(self._lib.spx_upload_peaks_pods(self._h, Table(self._hdr, "spx_peaks_pods_soa", n_pods=p, **pc).ref()))
        self.peaks_soa = dict(f["nodes"], cpu_milli=pc["cpu_milli"][:p])
        self.n_nodes, self.n_pods = f["N"], p

    def load_peaks_objects(self, nodes: Table, metrics: Table, power_models: Optional[Table], pods: Table) -> None:
        self.upload_peaks(self.flatten_peaks(nodes, metrics, power_models, pods))

    # ------------------------------------------------------------------ NodeResourceTopologyMatch
    def load_nrt_objects(self, nodes: Table, nrt: Table, rc: Optional[Table], pods: Table, params: Table) -> None:
        """objects -> (host flatten: slots, node zone tables, pod request tables) -> HBM."""
        self.upload_nrt(self.flatten_nrt(nodes, nrt, rc, pods, params))

    def flatten_nrt(self, nodes: Table, nrt: Table, rc: Optional[Table], pods: Table, params: Table) -> dict:
        L, H = self._lib, self._hdr
        u8p, i32p, i64p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_float)
        n_res = C.c_int32()
        slot_res = np.zeros(8, np.int32)
        slot_flags = np.zeros(8, np.uint8)
        slot_weight = np.zeros(8, np.int64)
        self._ck(L.spx_flatten_nrt_slots(pods.ref(), nrt.ref(), rc.ref() if rc else None, params.ref(), C.byref(n_res),
                                         slot_res.ctypes.data_as(i32p), slot_flags.ctypes.data_as(u8p),
                                         slot_weight.ctypes.data_as(i64p)))
        R = n_res.value
        slots = Table(H, "spx_nrt_slots", n_res=R, slot_res=slot_res, slot_flags=slot_flags, slot_weight=slot_weight)
        N, P = nodes.struct.n_nodes, pods.struct.n_pods
        nc = dict(flags=np.zeros(N, np.uint8), max_numa=np.zeros(N, np.int32), n_zones=np.zeros(N, np.uint8),
                  zone_id=np.zeros(N * 8, np.uint8), zone_present=np.zeros(N * 8, np.uint8),
                  zone_avail=np.zeros(N * 8 * max(R, 1), np.int64), zone_cost=np.zeros(N * 64, np.int32),
                  min_avg_dist=np.zeros(N * 8, np.float32), node_present=np.zeros(N, np.uint8))
        fn = L.spx_flatten_nrt_nodes
        self._ck(fn(nodes.ref(), nrt.ref(), slots.ref(), *[v.ctypes.data_as(t) for v, t in zip(nc.values(), fn.argtypes[3:])]))
        pc = dict(qos=np.zeros(P, np.uint8), non_native=np.zeros(P, np.uint8), n_ctr=np.zeros(P, np.uint8),
                  ctr_kind=np.zeros(P * 8, np.uint8), ctr_present=np.zeros(P * 8, np.uint8),
                  ctr_req=np.zeros(P * 8 * max(R, 1), np.int64), pod_present=np.zeros(P, np.uint8),
                  pod_req=np.zeros(P * max(R, 1), np.int64))
        fn = L.spx_flatten_nrt_pods
        self._ck(fn(pods.ref(), rc.ref() if rc else None, slots.ref(), *[v.ctypes.data_as(t) for v, t in zip(pc.values(), fn.argtypes[3:])]))
        return {"params": params, "slots": slots, "nodes": nc, "pods": pc, "N": N, "P": P, "R": R}

    def flatten_nrt_node_rows(self, nodes: Table, nrt: Table, slots: Table, idx) -> Dict[str, np.ndarray]:
        """the SoA rows of the listed nodes only (spx_flatten_nrt_node_rows): what a delta encoder produces for the changed nodes"""
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        n, R = len(idx), int(slots.struct.n_res)
        nc = dict(flags=np.zeros(n, np.uint8), max_numa=np.zeros(n, np.int32), n_zones=np.zeros(n, np.uint8),
                  zone_id=np.zeros(n * 8, np.uint8), zone_present=np.zeros(n * 8, np.uint8),
                  zone_avail=np.zeros(n * 8 * max(R, 1), np.int64), zone_cost=np.zeros(n * 64, np.int32),
                  min_avg_dist=np.zeros(n * 8, np.float32), node_present=np.zeros(n, np.uint8))
        fn = self._lib.spx_flatten_nrt_node_rows
        self._ck(fn(nodes.ref(), nrt.ref(), slots.ref(), idx.ctypes.data_as(C.POINTER(C.c_int64)), n,
                    *[v.ctypes.data_as(t) for v, t in zip(nc.values(), fn.argtypes[5:])]))
        return nc

    def update_nrt_node_rows(self, idx, rows: Dict[str, np.ndarray], n_res: int) -> None:
        """spx_update_nrt_nodes with rows as flatten_nrt_node_rows returns them"""
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        self._ck(self._lib.spx_update_nrt_nodes(self._h, idx.ctypes.data_as(C.POINTER(C.c_int64)),
                                                Table(self._hdr, "spx_nrt_nodes_soa", n_nodes=len(idx), n_res=n_res, **rows).ref()))

    def flatten_nrt_pods(self, pods: Table, rc: Optional[Table], slots: Table) -> Dict[str, np.ndarray]:
        """the pod half of flatten_nrt for a NEW pending batch against slots already uploaded (a cycle's pod delta)"""
        u8p, i64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int64)
        P, R = pods.struct.n_pods, int(slots.struct.n_res)
        pc = dict(qos=np.zeros(P, np.uint8), non_native=np.zeros(P, np.uint8), n_ctr=np.zeros(P, np.uint8),
                  ctr_kind=np.zeros(P * 8, np.uint8), ctr_present=np.zeros(P * 8, np.uint8),
                  ctr_req=np.zeros(P * 8 * max(R, 1), np.int64), pod_present=np.zeros(P, np.uint8),
                  pod_req=np.zeros(P * max(R, 1), np.int64))
        fn = self._lib.spx_flatten_nrt_pods
        self._ck(fn(pods.ref(), rc.ref() if rc else None, slots.ref(), *[v.ctypes.data_as(t) for v, t in zip(pc.values(), fn.argtypes[3:])]))
        return pc

    def upload_nrt_pods(self, pc: Dict[str, np.ndarray], n_res: int) -> None:
        P = len(pc["qos"])
        self._ck(self._lib.spx_upload_nrt_pods(self._h, Table(self._hdr, "spx_nrt_pods_soa", n_pods=P, n_res=n_res, **pc).ref()))
        self.n_pods = P
        self.nrt_soa["pods"] = pc

    def flatten_network_pods(self, pods: Table, appgroups: Table) -> dict:
        """the per-batch half of flatten_network (workload keys of the pending pods), without the commit effects"""
        L = self._lib
        i32p, i64p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
        P = pods.struct.n_pods
        nk, npairs = C.c_int32(), C.c_int64()
        self._ck(L.spx_flatten_net_keys(pods.ref(), appgroups.ref(), C.byref(nk), C.byref(npairs), None, None, None, None, None, None))
        cols = dict(pod_key=np.zeros(P, np.int32), topo_order=np.zeros(P, np.int32), key_score_equally=np.zeros(nk.value, np.uint8),
                    pair_ptr=np.zeros(nk.value + 1, np.int32), pair_node=np.zeros(max(npairs.value, 1), np.int32),
                    pair_max_cost=np.zeros(max(npairs.value, 1), np.int64))
        self._ck(L.spx_flatten_net_keys(pods.ref(), appgroups.ref(), C.byref(nk), C.byref(npairs),
                                        cols["pod_key"].ctypes.data_as(i32p), cols["topo_order"].ctypes.data_as(i32p),
                                        cols["key_score_equally"].ctypes.data_as(u8p), cols["pair_ptr"].ctypes.data_as(i32p),
                                        cols["pair_node"].ctypes.data_as(i32p), cols["pair_max_cost"].ctypes.data_as(i64p)))
        return {"cols": cols, "n_keys": nk.value, "P": P}

    def upload_network_pods(self, f: dict) -> None:
        self._ck(self._lib.spx_upload_net_pods(self._h, Table(self._hdr, "spx_net_pods_soa", n_pods=f["P"], n_keys=f["n_keys"], **f["cols"]).ref()))
        self.n_pods = f["P"]
        self.net_soa = f["cols"]

    def flatten_net_placed(self, pods: Table, appgroups: Table, group, selector, node) -> dict:
        """pods that joined AppGroup scheduled lists since flatten_network_pods(pods, appgroups): the entries update_net_placed appends"""
        L = self._lib
        i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        g, sel, nd = (np.ascontiguousarray(x, dtype=np.int32) for x in (group, selector, node))
        n = C.c_int64()
        args = (pods.ref(), appgroups.ref(), len(g), g.ctypes.data_as(i32p), sel.ctypes.data_as(i32p), nd.ctypes.data_as(i32p), C.byref(n))
        self._ck_static(L.spx_flatten_net_placed(*args, None, None, None))
        out = dict(key=np.zeros(max(n.value, 1), np.int32), node=np.zeros(max(n.value, 1), np.int32), max_cost=np.zeros(max(n.value, 1), np.int64))
        self._ck_static(L.spx_flatten_net_placed(*args, out["key"].ctypes.data_as(i32p), out["node"].ctypes.data_as(i32p), out["max_cost"].ctypes.data_as(i64p)))
        return {k: v[:n.value] for k, v in out.items()}

    def update_net_placed(self, ent: dict) -> None:
        """the workload keys' pair lists grown in place on the device (spx_update_net_placed)"""
        i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        k, nd, c = np.ascontiguousarray(ent["key"], np.int32), np.ascontiguousarray(ent["node"], np.int32), np.ascontiguousarray(ent["max_cost"], np.int64)
        self._ck(self._lib.spx_update_net_placed(self._h, len(k), k.ctypes.data_as(i32p), nd.ctypes.data_as(i32p), c.ctypes.data_as(i64p)))

    def update_quota_used(self, ns, used, used_present, agg_used, agg_used_present) -> None:
        """rows `ns` of ElasticQuotaInfo.Used replaced in place, with the new aggregate (spx_update_quota_used)"""
        i32p, i64p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
        ns = np.ascontiguousarray(ns, np.int32)
        used = np.ascontiguousarray(used, np.int64).reshape(-1)
        up, au, aup = np.ascontiguousarray(used_present, np.uint8), np.ascontiguousarray(agg_used, np.int64), np.ascontiguousarray(agg_used_present, np.uint8).reshape(-1)
        assert used.size == len(ns) * 8 and au.size == 8
        self._ck(self._lib.spx_update_quota_used(self._h, len(ns), ns.ctypes.data_as(i32p), used.ctypes.data_as(i64p), up.ctypes.data_as(u8p),
                                                 au.ctypes.data_as(i64p), aup.ctypes.data_as(u8p)))

    def upload_nrt(self, f: dict, rows=None) -> None:
        """rows = (begin, end): this engine holds only that slice of the pod batch (MultiEngine)"""
        L, H = self._lib, self._hdr
        self._ck(L.spx_set_nrt_params(self._h, f["params"].ref()))
        self._ck(L.spx_upload_nrt_slots(self._h, f["slots"].ref()))
        self._ck(L.spx_upload_nrt_nodes(self._h, Table(H, "spx_nrt_nodes_soa", n_nodes=f["N"], n_res=f["R"], **f["nodes"]).ref()))
        pc = _rows(f["pods"], f["P"], rows)
        P = f["P"] if rows is None else rows[1] - rows[0]
        if P > 0:
            self._ck(L.spx_upload_nrt_pods(self._h, Table(H, "spx_nrt_pods_soa", n_pods=P, n_res=f["R"], **pc).ref()))
        self.n_nodes, self.n_pods = f["N"], P
        self.nrt_soa = {"slots": f["slots"], "nodes": f["nodes"], "pods": pc}

    # ------------------------------------------------------------------ NetworkOverhead / TopologicalSort
    def load_network_objects(self, nodes: Table, pods: Table, appgroups: Table, nettopo: Table) -> None:
        self.upload_network(self.flatten_network(nodes, pods, appgroups, nettopo))

    def flatten_network(self, nodes: Table, pods: Table, appgroups: Table, nettopo: Table) -> dict:
        L, H = self._lib, self._hdr
        i32p, i64p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
        N, P = nodes.struct.n_nodes, pods.struct.n_pods
        rg, zc = nettopo.struct.n_regions, nettopo.struct.n_zones
        rcost = np.full(max(rg * rg, 1), -1, np.int32)
        zcost = np.full(max(zc * zc, 1), -1, np.int32)
        self._ck(L.spx_flatten_net_topo(nettopo.ref(), rcost.ctypes.data_as(i32p), zcost.ctypes.data_as(i32p)))
        nk, npairs = C.c_int32(), C.c_int64()
        self._ck(L.spx_flatten_net_keys(pods.ref(), appgroups.ref(), C.byref(nk), C.byref(npairs), None, None, None, None, None, None))
        cols = dict(pod_key=np.zeros(P, np.int32), topo_order=np.zeros(P, np.int32), key_score_equally=np.zeros(nk.value, np.uint8),
                    pair_ptr=np.zeros(nk.value + 1, np.int32), pair_node=np.zeros(max(npairs.value, 1), np.int32),
                    pair_max_cost=np.zeros(max(npairs.value, 1), np.int64))
        self._ck(L.spx_flatten_net_keys(pods.ref(), appgroups.ref(), C.byref(nk), C.byref(npairs),
                                        cols["pod_key"].ctypes.data_as(i32p), cols["topo_order"].ctypes.data_as(i32p),
                                        cols["key_score_equally"].ctypes.data_as(u8p), cols["pair_ptr"].ctypes.data_as(i32p),
                                        cols["pair_node"].ctypes.data_as(i32p), cols["pair_max_cost"].ctypes.data_as(i64p)))
        # what binding each pod adds to the AppGroup scheduled lists (sequential commit loop)
        n_eff = C.c_int64()
        self._ck(L.spx_flatten_net_commit(pods.ref(), appgroups.ref(), C.byref(n_eff), None, None, None))
        eff = dict(eff_ptr=np.zeros(P + 1, np.int32), eff_key=np.zeros(max(n_eff.value, 1), np.int32), eff_max_cost=np.zeros(max(n_eff.value, 1), np.int64))
        self._ck(L.spx_flatten_net_commit(pods.ref(), appgroups.ref(), C.byref(n_eff), eff["eff_ptr"].ctypes.data_as(i32p),
                                          eff["eff_key"].ctypes.data_as(i32p), eff["eff_max_cost"].ctypes.data_as(i64p)))
        return {"region": nodes.array("region"), "zone": nodes.array("zone"), "rg": rg, "zc": zc, "rcost": rcost, "zcost": zcost,
                "n_keys": nk.value, "cols": cols, "N": N, "P": P, "commit": eff}

    def upload_network(self, f: dict, rows=None) -> None:
        L, H = self._lib, self._hdr
        self._ck(L.spx_upload_net_nodes(self._h, Table(H, "spx_net_nodes_soa", n_nodes=f["N"], region=f["region"], zone=f["zone"]).ref()))
        self._ck(L.spx_upload_net_topo(self._h, Table(H, "spx_net_topo_soa", n_regions=f["rg"], n_zones=f["zc"], region_cost=f["rcost"],
                                                       zone_cost=f["zcost"]).ref()))
        cols = dict(f["cols"])
        P = f["P"] if rows is None else rows[1] - rows[0]
        cols.update(_rows({k: cols[k] for k in ("pod_key", "topo_order")}, f["P"], rows))  # the key tables are per workload, not per pod
        if P > 0:
            self._ck(L.spx_upload_net_pods(self._h, Table(H, "spx_net_pods_soa", n_pods=P, n_keys=f["n_keys"], **cols).ref()))
            if rows is None:  # the commit effects index the whole batch
                self._ck(L.spx_upload_net_commit(self._h, Table(H, "spx_net_commit_soa", n_pods=P, **f["commit"]).ref()))
        self.n_nodes, self.n_pods = f["N"], P
        self.net_soa = cols

    def sort_queue(self, pods: Table, topo_order: Optional[np.ndarray] = None) -> np.ndarray:
        """TopologicalSort as one device sort: queue order (pod rows) in which every adjacent pair satisfies Less (spx_sort_keys)"""
        topo = np.ascontiguousarray(self.net_soa["topo_order"] if topo_order is None else topo_order, dtype=np.int32)
        n = pods.struct.n_pods
        t = Table(self._hdr, "spx_sort_keys_soa", n_pods=n, priority=pods.array("priority"), queue_ts=pods.array("queue_ts"),
                  appgroup=pods.array("appgroup"), topo_order=topo)
        self._ck(self._lib.spx_upload_sort_keys(self._h, t.ref()))
        perm = np.zeros(n, np.int32)
        self._ck(self._lib.spx_sort_keys(self._h, perm.ctypes.data_as(C.POINTER(C.c_int32))))
        return perm

    def toposort_less(self, pods: Table, a: Sequence[int], b: Sequence[int]) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.int64)
        b = np.ascontiguousarray(b, dtype=np.int64)
        out = np.zeros(len(a), np.uint8)
        i64p = C.POINTER(C.c_int64)
        self._ck_static(self._lib.spx_toposort_less(pods.ref(), self.net_soa["topo_order"].ctypes.data_as(C.POINTER(C.c_int32)), len(a),
                                                    a.ctypes.data_as(i64p), b.ctypes.data_as(i64p),
                                                    out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out.astype(bool)

    @staticmethod
    def _ck_static(rc: int) -> None:
        if rc != 0:
            raise RuntimeError(f"spx host call failed: {rc}")

    # ------------------------------------------------------------------ CapacityScheduling.PreFilter
    def load_quota_objects(self, pods: Table, rc: Optional[Table], quota: Table) -> None:
        self.upload_quota(self.flatten_quota(pods, rc, quota))

    def flatten_quota(self, pods: Table, rc: Optional[Table], quota: Table) -> dict:
        L, H = self._lib, self._hdr
        P, NS = pods.struct.n_pods, quota.struct.n_namespaces
        nn = max(int(quota.struct.n_nominated), 1)
        cols = dict(pod_ns=np.zeros(P, np.int32), pod_priority=np.zeros(P, np.int32), pod_req=np.zeros(P * 8, np.int64),
                    pod_req_present=np.zeros(P, np.uint8), agg_used=np.zeros(8, np.int64), agg_used_present=np.zeros(1, np.uint8),
                    agg_min=np.zeros(8, np.int64), agg_min_present=np.zeros(1, np.uint8),
                    other_nominated=np.zeros(max(NS, 1) * 8, np.int64), other_nominated_present=np.zeros(max(NS, 1), np.uint8),
                    nom_ptr=np.zeros(NS + 1, np.int32), nom_priority=np.zeros(nn, np.int32), nom_pending_index=np.zeros(nn, np.int64),
                    nom_req=np.zeros(nn * 8, np.int64), nom_req_present=np.zeros(nn, np.uint8))
        fn = L.spx_flatten_quota
        self._ck_static(fn(pods.ref(), rc.ref() if rc else None, quota.ref(),
                           *[v.ctypes.data_as(t) for v, t in zip(cols.values(), fn.argtypes[3:])]))
        ns = dict(has_quota=quota.array("has_quota"), used=quota.array("used"), used_present=quota.array("used_present"),
                  max=quota.array("max"), max_present=quota.array("max_present"), min=quota.array("min"), min_present=quota.array("min_present"))
        return {"cols": cols, "ns": ns, "P": P, "NS": NS}

    _QUOTA_POD_COLS = ("pod_ns", "pod_priority", "pod_req", "pod_req_present")

    def upload_quota(self, f: dict, rows=None) -> None:
        cols = dict(f["cols"])
        P = f["P"] if rows is None else rows[1] - rows[0]
        cols.update(_rows({k: cols[k] for k in self._QUOTA_POD_COLS}, f["P"], rows))
        if rows is not None:  # a nominated pod is skipped when it is the pod under evaluation: indices are relative to this engine's rows
            cols["nom_pending_index"] = cols["nom_pending_index"] - rows[0]
        if P > 0:
            t = Table(self._hdr, "spx_quota_soa", n_pods=P, n_namespaces=f["NS"], **f["ns"], **cols)
            self._ck(self._lib.spx_upload_quota(self._h, t.ref()))
        self.n_pods = P

    def prefilter(self, plugin: int, row_begin: int = 0, row_end: Optional[int] = None) -> np.ndarray:
        row_end = self.n_pods if row_end is None else row_end
        out = np.zeros(row_end - row_begin, np.uint8)
        self._ck(self._lib.spx_fetch_prefilter(self._h, plugin, row_begin, row_end, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def status(self, plugin: int, pod_row: int) -> np.ndarray:
        out = np.empty(self.n_nodes, dtype=np.uint8)
        self._ck(self._lib.spx_fetch_status(self._h, plugin, pod_row, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def all_status(self, plugin: int, row_begin: int = 0, row_end: Optional[int] = None) -> np.ndarray:
        """[rows][n_nodes] uint8 in one strided copy (spx_fetch_status_rows)"""
        row_end = self.n_pods if row_end is None else row_end
        out = np.empty((row_end - row_begin, self.n_nodes), dtype=np.uint8)
        self._ck(self._lib.spx_fetch_status_rows(self._h, plugin, row_begin, row_end, out.ctypes.data_as(C.POINTER(C.c_uint8)), self.n_nodes))
        return out

    # ------------------------------------------------------------------ eval / fetch
    def eval(self, plugin_mask: int, row_begin: int = 0, row_end: Optional[int] = None) -> None:
        self._ck(self._lib.spx_eval(self._h, plugin_mask, row_begin, self.n_pods if row_end is None else row_end))

    def sync(self) -> None:
        self._ck(self._lib.spx_sync(self._h))

    def kernel_path(self, plugin: int) -> int:
        """0 = generic sweep kernel, 1 = fast formulation (same results)."""
        return int(self._lib.spx_kernel_path(self._h, plugin))

    # ------------------------------------------------------------------ one-call loaders (spx_load_*: flatten + upload inside the library)
    def load_c(self, snap: dict, nrt_params: Optional[Table] = None, concurrent: bool = False) -> None:
        """the object tables of `snap` (keys as synth.full_snapshot's) through spx_load_trimaran / _nrt / _network / _quota — the calls
        the cgo shim makes — instead of this module's own flatten_* + upload_* sequences"""
        L = self._lib
        ref = lambda t: t.ref() if t is not None else None
        if concurrent:  # spx_load_profile: the four loaders side by side inside the library
            keys = {"nodes": "nodes", "rc": "rc", "pods": "pods", "metrics": "metrics", "assigned": "assigned", "nrt": "nrt", "appgroups": "appgroups",
                    "nettopo": "nettopo", "quota": "quota"}
            fields = {k: snap[v] for k, v in keys.items() if snap.get(v) is not None}
            if "nrt" in fields:
                fields["nrt_params"] = nrt_params
            self._ck(L.spx_load_profile(self._h, Table(self._hdr, "spx_profile_objects", **fields).ref()))
            self.n_pods = snap["pods"].struct.n_pods
            self.n_nodes = snap["nodes"].struct.n_nodes
            return
        if "metrics" in snap:
            self._ck(L.spx_load_trimaran(self._h, snap["nodes"].ref(), ref(snap.get("rc")), snap["pods"].ref(), snap["metrics"].ref(), ref(snap.get("assigned"))))
        if "nrt" in snap:
            self._ck(L.spx_load_nrt(self._h, snap["nodes"].ref(), snap["nrt"].ref(), ref(snap.get("rc")), snap["pods"].ref(), nrt_params.ref()))
        if "appgroups" in snap:
            self._ck(L.spx_load_network(self._h, snap["nodes"].ref(), snap["pods"].ref(), snap["appgroups"].ref(), snap["nettopo"].ref()))
        if "quota" in snap:
            self._ck(L.spx_load_quota(self._h, snap["pods"].ref(), ref(snap.get("rc")), snap["quota"].ref()))
        self.n_pods = snap["pods"].struct.n_pods
        if "nodes" in snap:
            self.n_nodes = snap["nodes"].struct.n_nodes

    def last_load_nrt_ms(self) -> Dict[str, float]:
        """wall time of the stages of the last spx_load_nrt (spx_last_load_nrt_ms)"""
        ms = (C.c_double * 6)()
        self._ck(self._lib.spx_last_load_nrt_ms(self._h, ms))
        return dict(zip(("flatten_slots", "flatten_nodes", "flatten_pods", "params_slot_table", "upload_nodes", "upload_pods"), (float(x) for x in ms)))

    def load_trimaran_pods(self, pods: Table) -> None:
        """a new pending batch for Allocatable / TLP / LVRB: flattened straight into the engine's pinned staging (spx_load_trimaran_pods)"""
        self._ck(self._lib.spx_load_trimaran_pods(self._h, pods.ref()))
        self.n_pods = pods.struct.n_pods

    def nrt_packed_score_slots(self):
        """None when LeastAllocated's Score launch keeps float64; else (mask of the weighted NRT slots scored in packed float32
        unconditionally, the slot scored that way through the per-launch table or -1) — spx_nrt_packed_score_slots"""
        v = int(self._lib.spx_nrt_packed_score_slots(self._h))
        if v < 0:
            self._ck(v)
        if v == 0:
            return None
        return v & 0xffff, ((v >> 16) & 0xff) - 1

    def nrt_filter_path(self) -> int:
        """which Filter launch the last NRT sweep ran: 1 float64 compares, 2 rank space"""
        return int(self._lib.spx_nrt_filter_path(self._h))

    def commit_path(self) -> int:
        """which form the last commit_sequential ran: 1 one-workgroup trimaran chain, 2 per-pod launches, 3 cooperative kernel"""
        return int(self._lib.spx_commit_path(self._h))

    def last_eval_ms(self) -> float:
        ms = C.c_float()
        self._ck(self._lib.spx_last_eval_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def scores(self, plugin: int, pod_row: int) -> np.ndarray:
        out = np.empty(self.n_nodes, dtype=np.uint8)
        self._ck(self._lib.spx_fetch_scores(self._h, plugin, pod_row, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def raw(self, plugin: int, pod_row: int, which: int = 0) -> np.ndarray:
        out = np.empty(self.n_nodes, dtype=np.int64)
        self._ck(self._lib.spx_fetch_raw(self._h, plugin, which, pod_row, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def score_table(self, plugin: int):
        """(device pointer, row stride in bytes, rows) of a plugin's uint8 table in HBM."""
        p = C.c_void_p()
        stride = C.c_int64()
        rows = C.c_int64()
        self._ck(self._lib.spx_score_table(self._h, plugin, C.byref(p), C.byref(stride), C.byref(rows)))
        return p.value, stride.value, rows.value

    def bind_score_table(self, plugin: int, dptr: int, row_stride: int, n_rows: int) -> None:
        self._ck(self._lib.spx_bind_score_table(self._h, plugin, C.c_void_p(dptr), row_stride, n_rows))

    def upload_feasible_mask(self, mask: Optional[np.ndarray]) -> None:
        """[n_pods][n_nodes] uint8, non-zero = the node passed the caller's other Filter plugins; None clears it."""
        if mask is None:
            self._ck(self._lib.spx_upload_feasible_mask(self._h, None, 0, 0))
            return
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self._ck(self._lib.spx_upload_feasible_mask(self._h, mask.ctypes.data_as(C.POINTER(C.c_uint8)), mask.shape[0], mask.shape[1]))

    def set_plugin_weights(self, weights: Dict[int, int]) -> None:
        w = np.ones(NUM_PLUGINS, dtype=np.int64)
        for k, v in weights.items():
            w[k] = v
        self._ck(self._lib.spx_set_plugin_weights(self._h, w.ctypes.data_as(C.POINTER(C.c_int64))))

    def eval_best(self, plugin_mask: int, row_begin: int = 0, row_end: Optional[int] = None) -> None:
        self._ck(self._lib.spx_eval_best(self._h, plugin_mask, row_begin, self.n_pods if row_end is None else row_end))

    def decide(self, plugin_mask: int, row_begin: int = 0, row_end: Optional[int] = None) -> None:
        """eval + per-row argmax without materialising score tables where the profile allows (spx_decide); read with best()"""
        self._ck(self._lib.spx_decide(self._h, plugin_mask, row_begin, self.n_pods if row_end is None else row_end))

    def best(self, row_begin: int = 0, row_end: Optional[int] = None):
        """(best node, weighted score, ties, feasible count) per pod row."""
        row_end = self.n_pods if row_end is None else row_end
        n = row_end - row_begin
        node, score = np.zeros(n, np.int32), np.zeros(n, np.int64)
        ties, feas = np.zeros(n, np.int32), np.zeros(n, np.int32)
        i32p = C.POINTER(C.c_int32)
        self._ck(self._lib.spx_fetch_best(self._h, row_begin, row_end, node.ctypes.data_as(i32p), score.ctypes.data_as(C.POINTER(C.c_int64)),
                                          ties.ctypes.data_as(i32p), feas.ctypes.data_as(i32p)))
        return node, score, ties, feas

    def commit_sequential(self, plugin_mask: int, row_begin: int = 0, row_end: Optional[int] = None, want_ties: bool = True):
        """pods in row order, each seeing the commits before it -> (node, weighted score, ties, missing); plugin_mask may hold
        Allocatable / TLP / LVRB / NRT / NetworkOverhead / CapacityScheduling (spx_commit_sequential)"""
        row_end = self.n_pods if row_end is None else row_end
        n = row_end - row_begin
        node, score, ties = np.zeros(n, np.int32), np.zeros(n, np.int64), np.zeros(n, np.int32)
        missing = np.zeros(self.n_nodes, np.int64)
        self._ck(self._lib.spx_commit_sequential(self._h, plugin_mask, row_begin, row_end, node.ctypes.data_as(C.POINTER(C.c_int32)),
                                                 score.ctypes.data_as(C.POINTER(C.c_int64)),
                                                 ties.ctypes.data_as(C.POINTER(C.c_int32)) if want_ties else None,
                                                 missing.ctypes.data_as(C.POINTER(C.c_int64))))
        return node, score, (ties if want_ties else None), missing

    def set_stream(self, stream: int) -> None:
        self._ck(self._lib.spx_set_stream(self._h, C.c_void_p(stream)))

    def all_scores(self, plugin: int, row_begin: int = 0, row_end: Optional[int] = None) -> np.ndarray:
        """[rows][n_nodes] uint8 in one strided copy (spx_fetch_score_rows)"""
        row_end = self.n_pods if row_end is None else row_end
        out = np.empty((row_end - row_begin, self.n_nodes), dtype=np.uint8)
        self._ck(self._lib.spx_fetch_score_rows(self._h, plugin, row_begin, row_end, out.ctypes.data_as(C.POINTER(C.c_uint8)), self.n_nodes))
        return out
