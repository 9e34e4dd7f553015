"""Shared builders for the table-driven tests (mirrors the reference tests' makeNodeInfo / st.MakePod)."""
from __future__ import annotations

import numpy as np

import scheduler_plugins_amd as spx
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd._abi import Table

ALLOCATABLE, TLP, LVRB, NRT, NETOVERHEAD, CAPACITY, TOPOSORT, LROC, PEAKS = range(9)


def alloc_params(hdr, res: O.Resources, resources: dict, mode: str) -> Table:
    ids = [res.id(k) for k in resources]
    return Table(hdr, "spx_allocatable_params", mode={"Least": 0, "Most": 1}[mode], n_res=len(ids),
                 res=np.array(ids, dtype=np.int32), weight=np.array(list(resources.values()), dtype=np.int64))


def tlp_params(hdr, target_utilization=40, default_requests_milli=1000, requests_multiplier=1.5) -> Table:
    return Table(hdr, "spx_tlp_params", target_utilization=target_utilization,
                 default_requests_milli=default_requests_milli, requests_multiplier=requests_multiplier)


def lvrb_params(hdr, margin=1.0, sensitivity=1.0) -> Table:
    return Table(hdr, "spx_lvrb_params", safe_variance_margin=margin, safe_variance_sensitivity=sensitivity)


def lroc_params(hdr, smoothing_window_size=5, w_cpu=0.5, w_mem=0.5) -> Table:
    return Table(hdr, "spx_lroc_params", smoothing_window_size=smoothing_window_size, risk_limit_weight_cpu=w_cpu, risk_limit_weight_mem=w_mem)


def power_models(hdr, models) -> Table:
    """models: one dict {"k0","k1","k2"} or None per node (getPowerModel returns the zero model for an unknown node)"""
    g = lambda k: np.array([(m or {}).get(k, 0.0) for m in models], dtype=np.float64)
    return Table(hdr, "spx_power_model_objects", k0=g("k0"), k1=g("k1"), k2=g("k2"))


def make_node_info(milli_cpu: int, memory: int) -> dict:  # allocatable_test.go:315-331
    return O.node({"cpu": f"{milli_cpu}m", "memory": memory})
