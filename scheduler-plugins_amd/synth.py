"""Seeded synthetic cluster snapshots as object tables (numpy, vectorised, any scale).

Value distributions follow SURVEY.md §8d: node sizes mirror the reference benchmarks' 64-core
nodes (pkg/trimaran/targetloadpacking/targetloadpacking_test.go:452-456), pods are shaped like
the reference's test pods (1-3 app containers, optional init container, QoS mix).  The same
arrays feed the CPU oracle and, through the host flatteners, the GPU — that is the parity setup.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from ._abi import Header, Table

SEED = 20260921
GiB = 1 << 30
MiB = 1 << 20


def _csr_from_mask(mask: np.ndarray):
    """mask [n, slots] -> (ptr [n+1], flat selector) keeping row-major slot order."""
    counts = mask.sum(axis=1)
    ptr = np.zeros(mask.shape[0] + 1, dtype=np.int32)
    np.cumsum(counts, out=ptr[1:])
    return ptr, mask.reshape(-1)


def synth_pods(hdr: Header, n_pods: int, seed: int = SEED, device_res: int = -1, n_appgroups: int = 0,
               n_namespaces: int = 100) -> Table:
    rng = np.random.default_rng(seed + 1)
    n_app = rng.integers(1, 4, n_pods)
    has_init = rng.random(n_pods) < 0.2
    n_ctr = n_app + has_init
    ctr_ptr = np.zeros(n_pods + 1, dtype=np.int32)
    np.cumsum(n_ctr, out=ctr_ptr[1:])
    total = int(ctr_ptr[-1])
    pod_of = np.repeat(np.arange(n_pods), n_ctr)
    pos = np.arange(total) - ctr_ptr[pod_of]
    kind = np.where(has_init[pod_of] & (pos == 0), 1, 0).astype(np.uint8)
    # 10% of init containers are sidecars
    kind = np.where((kind == 1) & (rng.random(total) < 0.1), 2, kind).astype(np.uint8)

    qos = rng.choice(3, n_pods, p=[0.5, 0.4, 0.1])  # 0 Guaranteed, 1 Burstable, 2 BestEffort
    q = qos[pod_of]
    cpu = np.exp(rng.uniform(np.log(100), np.log(8000), total)).astype(np.int64)
    whole = rng.random(total) < 0.3
    cpu = np.where(whole, np.maximum(1, (cpu + 500) // 1000) * 1000, cpu)
    mem = (np.exp(rng.uniform(np.log(64), np.log(32 * 1024), total)).astype(np.int64)) * MiB
    dev = (rng.random(n_pods) < 0.1)[pod_of] & (device_res >= 0) & (kind == 0) & (pos == has_init[pod_of])
    dev_n = rng.integers(1, 3, total)

    # requests: slots (cpu, memory, device)
    bur_cpu = rng.random(total) < 0.8  # burstable containers may omit one of the two
    bur_mem = rng.random(total) < 0.8
    req_mask = np.stack([(q == 0) | ((q == 1) & bur_cpu), (q == 0) | ((q == 1) & bur_mem), dev], axis=1)
    req_res = np.tile(np.array([0, 1, max(device_res, 0)], dtype=np.int32), (total, 1))
    req_qty = np.stack([cpu, mem, dev_n], axis=1)
    req_ptr, sel = _csr_from_mask(req_mask)
    # limits: Guaranteed == requests; Burstable sometimes a larger cpu limit; devices always limit == request
    bur_lim = (q == 1) & bur_cpu & (rng.random(total) < 0.5)
    lim_mask = np.stack([(q == 0) | bur_lim, (q == 0), dev], axis=1)
    lim_qty = np.stack([np.where(q == 0, cpu, cpu * 2), mem, dev_n], axis=1)
    lim_ptr, lsel = _csr_from_mask(lim_mask)

    ovh = rng.random(n_pods) < 0.05
    ovh_mask = np.stack([ovh, ovh & (rng.random(n_pods) < 0.5)], axis=1)
    ovh_qty = np.stack([rng.integers(50, 500, n_pods), rng.integers(16, 256, n_pods) * MiB], axis=1)
    ovh_res = np.tile(np.array([0, 1], dtype=np.int32), (n_pods, 1))
    ovh_ptr, osel = _csr_from_mask(ovh_mask)

    if n_appgroups > 0:
        appgroup = np.where(rng.random(n_pods) < 0.9, rng.integers(0, n_appgroups, n_pods), -1).astype(np.int32)
        selector = np.where(appgroup >= 0, rng.integers(0, 11, n_pods), -1).astype(np.int32)
    else:
        appgroup = np.full(n_pods, -1, dtype=np.int32)
        selector = np.full(n_pods, -1, dtype=np.int32)
    return Table(
        hdr, "spx_pod_objects", n_pods=n_pods, ctr_ptr=ctr_ptr, ctr_kind=kind,
        req_ptr=req_ptr, req_res=req_res.reshape(-1)[sel], req_qty=req_qty.reshape(-1)[sel],
        lim_ptr=lim_ptr, lim_res=req_res.reshape(-1)[lsel], lim_qty=lim_qty.reshape(-1)[lsel],
        ovh_ptr=ovh_ptr, ovh_res=ovh_res.reshape(-1)[osel], ovh_qty=ovh_qty.reshape(-1)[osel],
        priority=rng.choice(np.array([0, 100, 1000], dtype=np.int32), n_pods),
        queue_ts=np.arange(n_pods, dtype=np.int64) * 1000 + 1_700_000_000_000_000,
        appgroup=appgroup, selector=selector, ns=rng.integers(0, n_namespaces, n_pods).astype(np.int32),
    )


def synth_nodes(hdr: Header, n_nodes: int, seed: int = SEED, device_res: int = -1, n_regions: int = 8,
                zones_per_region: int = 8) -> Table:
    rng = np.random.default_rng(seed + 2)
    cpu = rng.choice(np.array([8, 16, 32, 64, 96, 128], dtype=np.int64), n_nodes) * 1000
    mem = rng.choice(np.array([32, 64, 128, 256, 346, 512, 1024], dtype=np.int64), n_nodes) * GiB
    reserved = rng.integers(500, 2001, n_nodes)
    has_dev = (rng.random(n_nodes) < 0.5) & (device_res >= 0)
    sc_mask = has_dev.reshape(-1, 1)
    sc_ptr, ssel = _csr_from_mask(sc_mask)
    region = rng.integers(0, n_regions, n_nodes).astype(np.int32)
    zone = (region * zones_per_region + rng.integers(0, zones_per_region, n_nodes)).astype(np.int32)
    unl = rng.random(n_nodes) < 0.01
    region = np.where(unl, -1, region).astype(np.int32)
    zone = np.where(unl, -1, zone).astype(np.int32)
    return Table(
        hdr, "spx_node_objects", n_nodes=n_nodes,
        alloc_cpu_milli=cpu - reserved, alloc_mem=mem - 2 * GiB, alloc_eph=np.full(n_nodes, 500 * GiB, dtype=np.int64),
        alloc_pods=np.full(n_nodes, 110, dtype=np.int64),
        scalar_ptr=sc_ptr, scalar_res=np.full(n_nodes, max(device_res, 0), dtype=np.int32)[ssel],
        scalar_qty=rng.integers(1, 9, n_nodes)[ssel],
        cap_cpu_milli=cpu, region=region, zone=zone,
    )


def synth_metrics(hdr: Header, n_nodes: int, seed: int = SEED, window_end: int = 1_700_000_000,
                  round_frac: float = 0.0) -> Table:
    """Per node up to 6 metric slots in a fixed order that exercises SURVEY appendix B.5:
    [CPU AVG, CPU STD, CPU Latest, Memory AVG, Memory STD, Memory ""].  TLP takes the LAST of
    CPU AVG/Latest, LVRB prefers AVG regardless of order."""
    rng = np.random.default_rng(seed + 3)
    present = rng.random(n_nodes) >= 0.02
    nil = present & (rng.random(n_nodes) < 0.005)
    r = rng.random((n_nodes, 6))
    mask = np.stack([r[:, 0] < 0.9, r[:, 1] < 0.8, r[:, 2] < 0.3, r[:, 3] < 0.85, r[:, 4] < 0.7, r[:, 5] < 0.2], axis=1)
    mask &= (present & ~nil).reshape(-1, 1)
    mtype = np.tile(np.array([0, 0, 0, 1, 1, 1], dtype=np.uint8), (n_nodes, 1))
    mop = np.tile(np.array([0, 1, 2, 0, 1, 3], dtype=np.uint8), (n_nodes, 1))
    val = np.stack([rng.uniform(0, 100, n_nodes), rng.uniform(0, 30, n_nodes), rng.uniform(0, 100, n_nodes),
                    rng.uniform(0, 100, n_nodes), rng.uniform(0, 30, n_nodes), rng.uniform(0, 100, n_nodes)], axis=1)
    # optional slice of round values that land exactly on rounding ties / threshold boundaries (tests);
    # SURVEY.md §8d's distribution itself is continuous: cpu_util% ~ U[0,100)
    rnd = rng.random(n_nodes) < round_frac
    val = np.where(rnd.reshape(-1, 1), np.round(val), val)
    ptr, sel = _csr_from_mask(mask)
    return Table(hdr, "spx_metrics_objects", map_is_nil=0, window_end=window_end,
                 node_present=present.astype(np.uint8), node_metrics_nil=nil.astype(np.uint8), m_ptr=ptr,
                 m_type=mtype.reshape(-1)[sel], m_op=mop.reshape(-1)[sel], m_value=val.reshape(-1)[sel])


def synth_assigned(hdr: Header, n_nodes: int, seed: int = SEED, window_end: int = 1_700_000_000) -> Table:
    """ScheduledPodsCache: 10% of nodes hold 1-3 recently bound pods with timestamps straddling the
    metrics window end (both sides of the 60 s rule, targetloadpacking.go:158-159)."""
    rng = np.random.default_rng(seed + 4)
    cnt = np.where(rng.random(n_nodes) < 0.1, rng.integers(1, 4, n_nodes), 0)
    e_ptr = np.zeros(n_nodes + 1, dtype=np.int32)
    np.cumsum(cnt, out=e_ptr[1:])
    n_e = int(e_ptr[-1])
    ts = window_end + rng.integers(-200, 100, n_e)
    pods = synth_pods(hdr, max(n_e, 1), seed=seed + 77)
    return Table(hdr, "spx_assigned_objects", e_ptr=e_ptr, e_ts_unix=ts.astype(np.int64),
                 e_pod=np.arange(n_e, dtype=np.int32), pods=pods)


def resource_classes(hdr: Header, flags: Optional[np.ndarray] = None) -> Table:
    if flags is None:
        flags = np.zeros(8, dtype=np.uint8)
        flags[[0, 1, 2, 3, 4]] = 2  # native
    return Table(hdr, "spx_resource_classes", n_res=len(flags), flags=flags)


def trimaran_snapshot(hdr: Header, n_nodes: int, n_pods: int, seed: int = SEED, round_frac: float = 0.0) -> Dict[str, Table]:
    """Object tables for BASELINE.json config #2 (Allocatable + TargetLoadPacking [+ LVRB])."""
    return {
        "nodes": synth_nodes(hdr, n_nodes, seed),
        "pods": synth_pods(hdr, n_pods, seed),
        "metrics": synth_metrics(hdr, n_nodes, seed, round_frac=round_frac),
        "assigned": synth_assigned(hdr, n_nodes, seed),
        "rc": resource_classes(hdr),
    }
