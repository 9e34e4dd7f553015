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
               n_namespaces: int = 100, hugepage_res: int = -1, qos_p=(0.5, 0.4, 0.1), device2_res: int = -1, hugepage2_res: int = -1) -> Table:
    """`qos_p`: shares of Guaranteed / Burstable / BestEffort pods (50 / 40 / 10 %; experiments vary it).
    `device2_res` / `hugepage2_res`: a second extended resource and a second hugepage size (the six-slot NRT workload); their
    random streams are separate, so every other column is the same with or without them."""
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

    qos = rng.choice(3, n_pods, p=list(qos_p))  # 0 Guaranteed, 1 Burstable, 2 BestEffort
    q = qos[pod_of]
    cpu = np.exp(rng.uniform(np.log(100), np.log(8000), total)).astype(np.int64)
    whole = rng.random(total) < 0.3
    cpu = np.where(whole, np.maximum(1, (cpu + 500) // 1000) * 1000, cpu)
    mem = (np.exp(rng.uniform(np.log(64), np.log(32 * 1024), total)).astype(np.int64)) * MiB
    dev = (rng.random(n_pods) < 0.1)[pod_of] & (device_res >= 0) & (kind == 0) & (pos == has_init[pod_of])
    dev_n = rng.integers(1, 3, total)

    # requests: slots (cpu, memory, device, hugepages)
    rng_hp = np.random.default_rng(seed + 101)  # separate stream: keeps the other columns identical with/without hugepages
    hp = (rng_hp.random(total) < 0.15) & (hugepage_res >= 0) & (q != 2)
    hp_q = rng_hp.integers(0, 65, total).astype(np.int64) * (2 << 20)  # includes explicit zero-quantity requests
    bur_cpu = rng.random(total) < 0.8  # burstable containers may omit one of the two
    bur_mem = rng.random(total) < 0.8
    rng_w = np.random.default_rng(seed + 202)
    dev2 = (rng_w.random(n_pods) < 0.08)[pod_of] & (device2_res >= 0) & (kind == 0) & (pos == has_init[pod_of])
    dev2_n = rng_w.integers(1, 3, total)
    hp2 = (rng_w.random(total) < 0.08) & (hugepage2_res >= 0) & (q != 2)
    hp2_q = rng_w.integers(1, 5, total).astype(np.int64) * GiB
    req_mask = np.stack([(q == 0) | ((q == 1) & bur_cpu), (q == 0) | ((q == 1) & bur_mem), dev, hp, dev2, hp2], axis=1)
    req_res = np.tile(np.array([0, 1, max(device_res, 0), max(hugepage_res, 0), max(device2_res, 0), max(hugepage2_res, 0)], dtype=np.int32), (total, 1))
    req_qty = np.stack([cpu, mem, dev_n, hp_q, dev2_n, hp2_q], axis=1)
    req_ptr, sel = _csr_from_mask(req_mask)
    # limits: Guaranteed == requests; Burstable sometimes a larger cpu limit; devices/hugepages always limit == request
    bur_lim = (q == 1) & bur_cpu & (rng.random(total) < 0.5)
    lim_mask = np.stack([(q == 0) | bur_lim, (q == 0), dev, hp, dev2, hp2], axis=1)
    lim_qty = np.stack([np.where(q == 0, cpu, cpu * 2), mem, dev_n, hp_q, dev2_n, hp2_q], axis=1)
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



def _gather_csr(ptr: np.ndarray, idx: np.ndarray):
    """rows `idx` of a CSR structure: (new ptr, positions of the gathered entries in the old value arrays)"""
    cnt = (ptr[1:] - ptr[:-1])[idx]
    nptr = np.zeros(len(idx) + 1, dtype=np.int32)
    np.cumsum(cnt, out=nptr[1:])
    pos = np.repeat(ptr[:-1][idx] - nptr[:-1], cnt) + np.arange(int(nptr[-1]))
    return nptr, pos


def take_pods(hdr: Header, pods: Table, idx) -> Table:
    """The pod table made of rows `idx` of `pods`, in that order, repeats allowed: a queue of Deployment replicas is
    take_pods(templates, rng.integers(0, n_templates, n_pods)).  Queue timestamps are renumbered (they are per queue entry)."""
    idx = np.asarray(idx, dtype=np.int64)
    ctr_ptr, cpos = _gather_csr(pods.array("ctr_ptr"), idx)
    req_ptr, rpos = _gather_csr(pods.array("req_ptr"), cpos)
    lim_ptr, lpos = _gather_csr(pods.array("lim_ptr"), cpos)
    ovh_ptr, opos = _gather_csr(pods.array("ovh_ptr"), idx)
    return Table(
        hdr, "spx_pod_objects", n_pods=len(idx), ctr_ptr=ctr_ptr, ctr_kind=pods.array("ctr_kind")[cpos],
        req_ptr=req_ptr, req_res=pods.array("req_res")[rpos], req_qty=pods.array("req_qty")[rpos],
        lim_ptr=lim_ptr, lim_res=pods.array("lim_res")[lpos], lim_qty=pods.array("lim_qty")[lpos],
        ovh_ptr=ovh_ptr, ovh_res=pods.array("ovh_res")[opos], ovh_qty=pods.array("ovh_qty")[opos],
        priority=pods.array("priority")[idx], queue_ts=np.arange(len(idx), dtype=np.int64) * 1000 + 1_700_000_000_000_000,
        appgroup=pods.array("appgroup")[idx], selector=pods.array("selector")[idx], ns=pods.array("ns")[idx],
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


def synth_node_pods(hdr: Header, n_nodes: int, seed: int = SEED, max_pods: int = 12) -> Table:
    """framework.NodeInfo.GetPods(): 0..max_pods-1 pods already running per node, drawn like the pending pods (so a
    share of Burstable pods carries cpu limits of twice the request: small nodes end up over-committed on limits, large
    ones do not — both sides of lowriskovercommitment.go:206)."""
    rng = np.random.default_rng(seed + 5)
    cnt = rng.integers(0, max_pods, n_nodes)
    p_ptr = np.zeros(n_nodes + 1, dtype=np.int32)
    np.cumsum(cnt, out=p_ptr[1:])
    n_p = int(p_ptr[-1])
    return Table(hdr, "spx_node_pods_objects", p_ptr=p_ptr, p_pod=np.arange(n_p, dtype=np.int32),
                 pods=synth_pods(hdr, max(n_p, 1), seed=seed + 55))


def synth_power_models(hdr: Header, n_nodes: int, seed: int = SEED) -> Table:
    """PeaksArgs.NodePowerModel: Power = K0 + K1 * e^(K2 * utilisation) with K1, K2 < 0 around the reference's fixture
    (peaks_test.go:80-86); 5% of the nodes have no entry (getPowerModel then yields the zero model, peaks.go:190-196)."""
    rng = np.random.default_rng(seed + 6)
    has = rng.random(n_nodes) >= 0.05
    k0 = np.where(has, rng.uniform(300, 600, n_nodes), 0.0)
    k1 = np.where(has, -rng.uniform(40, 160, n_nodes), 0.0)
    k2 = np.where(has, -rng.uniform(0.02, 0.12, n_nodes), 0.0)
    return Table(hdr, "spx_power_model_objects", k0=k0, k1=k1, k2=k2)


def resource_classes(hdr: Header, flags: Optional[np.ndarray] = None) -> Table:
    if flags is None:
        flags = np.zeros(8, dtype=np.uint8)
        flags[[0, 1, 2, 3, 4]] = 2  # native
    return Table(hdr, "spx_resource_classes", n_res=len(flags), flags=flags)


def trimaran_snapshot(hdr: Header, n_nodes: int, n_pods: int, seed: int = SEED, round_frac: float = 0.0,
                      with_node_pods: bool = False) -> Dict[str, Table]:
    """Object tables for BASELINE.json config #2 (Allocatable + TargetLoadPacking [+ LVRB]); with_node_pods adds the pods
    already running on every node, which LowRiskOverCommitment sums."""
    snap = {
        "nodes": synth_nodes(hdr, n_nodes, seed),
        "pods": synth_pods(hdr, n_pods, seed),
        "metrics": synth_metrics(hdr, n_nodes, seed, round_frac=round_frac),
        "assigned": synth_assigned(hdr, n_nodes, seed),
        "rc": resource_classes(hdr),
    }
    if with_node_pods:
        snap["node_pods"] = synth_node_pods(hdr, n_nodes, seed)
    return snap


# ------------------------------------------------------------------ NodeResourceTopology (config #3)
RES_HUGEPAGES_2MI = 8   # "hugepages-2Mi"
RES_DEVICE = 9          # "example.com/gpu" style extended resource
RES_HUGEPAGES_1GI = 10  # "hugepages-1Gi"      (the six-slot workload: a cluster with two hugepage sizes and two device types)
RES_DEVICE2 = 11        # a second extended resource


def nrt_resource_classes(hdr: Header, wide: bool = False) -> Table:
    flags = np.zeros(12 if wide else 10, dtype=np.uint8)
    flags[[0, 1, 2, 3, 4]] = 2              # native
    flags[RES_HUGEPAGES_2MI] = 1 | 2 | 4    # hugepage, native, scalar
    flags[RES_DEVICE] = 4                   # extended: not native, scalar
    if wide:
        flags[RES_HUGEPAGES_1GI] = 1 | 2 | 4
        flags[RES_DEVICE2] = 4
    return Table(hdr, "spx_resource_classes", n_res=len(flags), flags=flags)


def synth_nrt(hdr: Header, nodes: Table, seed: int = SEED, n_zones: int = 8, vary: bool = True, wide: bool = False):
    """NRT objects for the given nodes: Z NUMA zones per node with ids == list positions,
    per-zone available = alloc/Z x U[0.1,1] for {cpu, memory, hugepages-2Mi, device}; distances 10 on the
    diagonal and {12,20,32} off it; single-numa-node policy, container scope 70% / pod scope 30%,
    1% of nodes without NRT, 1% stale.  `vary` adds the ragged cases the reference tests cover: fewer zones,
    zones that do not report the device, missing cost entries, non-single-numa policies, assumed pods."""
    rng = np.random.default_rng(seed + 5)
    N = nodes.struct.n_nodes
    cpu = nodes.array("alloc_cpu_milli")
    mem = nodes.array("alloc_mem")
    has = rng.random(N) >= 0.01
    fresh = rng.random(N) >= 0.01
    nz = np.full(N, n_zones)
    if vary:
        nz = np.where(rng.random(N) < 0.15, rng.choice(np.array([1, 2, 4]), N), nz)
    nz = np.where(has, nz, 0)
    zone_ptr = np.zeros(N + 1, dtype=np.int32)
    np.cumsum(nz, out=zone_ptr[1:])
    nzt = int(zone_ptr[-1])
    node_of = np.repeat(np.arange(N), nz)
    zpos = np.arange(nzt) - zone_ptr[node_of]
    frac = rng.uniform(0.1, 1.0, (nzt, 4))
    z_cpu = (cpu[node_of] / nz[node_of] * frac[:, 0]).astype(np.int64)
    whole = rng.random(nzt) < 0.5
    z_cpu = np.where(whole, (z_cpu // 1000) * 1000, z_cpu)
    z_mem = (mem[node_of] / nz[node_of] * frac[:, 1]).astype(np.int64)
    z_hp = (rng.integers(0, 513, nzt) * (2 << 20)).astype(np.int64)
    z_dev = rng.integers(0, 5, nzt).astype(np.int64)
    node_has_dev = rng.random(N) < 0.6
    rep_dev = node_has_dev[node_of] & ((rng.random(nzt) < 0.85) if vary else True)
    rep_hp = (rng.random(N) < 0.8)[node_of]
    mask = np.stack([np.ones(nzt, bool), np.ones(nzt, bool), rep_hp, rep_dev], axis=1)
    res = np.tile(np.array([0, 1, RES_HUGEPAGES_2MI, RES_DEVICE], dtype=np.int32), (nzt, 1))
    qty = np.stack([z_cpu, z_mem, z_hp, z_dev], axis=1)
    if wide:  # two more zone resources, from their own stream (the four above stay as they are)
        rng_w = np.random.default_rng(seed + 205)
        z_hp1g = (rng_w.integers(0, 9, nzt) * GiB).astype(np.int64)
        z_dev2 = rng_w.integers(0, 3, nzt).astype(np.int64)
        rep_hp1g = (rng_w.random(N) < 0.7)[node_of]
        rep_dev2 = (rng_w.random(N) < 0.5)[node_of] & (rng_w.random(nzt) < 0.9)
        mask = np.concatenate([mask, np.stack([rep_hp1g, rep_dev2], axis=1)], axis=1)
        res = np.tile(np.array([0, 1, RES_HUGEPAGES_2MI, RES_DEVICE, RES_HUGEPAGES_1GI, RES_DEVICE2], dtype=np.int32), (nzt, 1))
        qty = np.concatenate([qty, np.stack([z_hp1g, z_dev2], axis=1)], axis=1)
    zres_ptr, sel = _csr_from_mask(mask)
    # costs: full matrix per node, 5% of entries dropped when vary
    cnt = nz[node_of]
    cost_ptr = np.zeros(nzt + 1, dtype=np.int32)
    cmask = np.arange(8)[None, :] < cnt[:, None]
    if vary:
        cmask &= rng.random((nzt, 8)) >= 0.05
    tgt = np.tile(np.arange(8, dtype=np.int32), (nzt, 1))
    off = rng.choice(np.array([12, 20, 32]), (N, 8, 8))
    off = np.minimum(off, off.transpose(0, 2, 1))
    cval = np.where(tgt == zpos[:, None], 10, off[node_of[:, None], np.minimum(zpos, 7)[:, None], tgt])
    np.cumsum(cmask.sum(axis=1), out=cost_ptr[1:])
    csel = cmask.reshape(-1)
    # policies
    pod_scope = rng.random(N) < 0.3
    legacy = np.where(pod_scope, (3 << 1) | 1, (3 << 1) | 0).astype(np.int8)
    attr_scope = np.full(N, -1, dtype=np.int8)
    attr_policy = np.full(N, -1, dtype=np.int8)
    attr_max = np.full(N, -1, dtype=np.int32)
    if vary:
        other = rng.random(N) < 0.06
        legacy = np.where(other, rng.choice(np.array([-1, (1 << 1) | 0, (2 << 1) | 1], dtype=np.int8), N), legacy).astype(np.int8)
        via_attr = rng.random(N) < 0.1   # attributes override the legacy field
        attr_policy = np.where(via_attr, 3, attr_policy).astype(np.int8)
        attr_scope = np.where(via_attr, rng.integers(0, 2, N), attr_scope).astype(np.int8)
        attr_max = np.where(rng.random(N) < 0.1, rng.choice(np.array([4, 8, 16]), N), attr_max).astype(np.int32)
    # assumed pods (OverReserve): 5% of nodes carry 1-2 assumed pods
    acnt = np.where((rng.random(N) < 0.05) & has & vary, rng.integers(1, 3, N), 0)
    assumed_ptr = np.zeros(N + 1, dtype=np.int32)
    np.cumsum(acnt, out=assumed_ptr[1:])
    na = int(assumed_ptr[-1])
    arl_ptr = np.arange(na + 1, dtype=np.int32) * 2
    arl_res = np.tile(np.array([0, 1], dtype=np.int32), na)
    arl_qty = np.stack([rng.integers(500, 4000, na), rng.integers(1, 16, na) * GiB], axis=1).reshape(-1)
    return Table(
        hdr, "spx_nrt_objects", n_nodes=N, has_nrt=has.astype(np.uint8), fresh=fresh.astype(np.uint8),
        legacy_policy=legacy, attr_scope=attr_scope, attr_policy=attr_policy, attr_max_numa=attr_max,
        zone_ptr=zone_ptr, zone_is_node=np.ones(nzt, dtype=np.uint8), zone_numa_id=zpos.astype(np.int32),
        zres_ptr=zres_ptr, zres_res=res.reshape(-1)[sel], zres_avail=qty.reshape(-1)[sel],
        zcost_ptr=cost_ptr, zcost_numa_id=tgt.reshape(-1)[csel], zcost_value=cval.reshape(-1)[csel].astype(np.int64),
        assumed_ptr=assumed_ptr, arl_ptr=arl_ptr, arl_res=arl_res, arl_qty=arl_qty.astype(np.int64),
    )


def nrt_snapshot(hdr: Header, n_nodes: int, n_pods: int, seed: int = SEED, vary: bool = True, wide: bool = False) -> Dict[str, Table]:
    """Object tables for BASELINE.json config #3 (NRT Filter+Score, 8 NUMA zones).  `wide`: six resource slots (cpu, memory,
    hugepages-2Mi, hugepages-1Gi, two extended resources) instead of four — the kernels' 8-slot instantiations."""
    nodes = synth_nodes(hdr, n_nodes, seed, device_res=RES_DEVICE)
    # node-level allocatable must list hugepages too (util.ResourceList key check, filter.go:110-116)
    rng = np.random.default_rng(seed + 6)
    N = n_nodes
    has_hp = rng.random(N) < 0.9
    has_dev = rng.random(N) < 0.7
    mask = np.stack([has_hp, has_dev], axis=1)
    res = np.tile(np.array([RES_HUGEPAGES_2MI, RES_DEVICE], dtype=np.int32), (N, 1))
    qty = np.stack([np.full(N, 1 << 30, dtype=np.int64), rng.integers(1, 17, N)], axis=1)
    if wide:
        rng_w = np.random.default_rng(seed + 206)
        mask = np.concatenate([mask, np.stack([rng_w.random(N) < 0.85, rng_w.random(N) < 0.6], axis=1)], axis=1)
        res = np.tile(np.array([RES_HUGEPAGES_2MI, RES_DEVICE, RES_HUGEPAGES_1GI, RES_DEVICE2], dtype=np.int32), (N, 1))
        qty = np.concatenate([qty, np.stack([np.full(N, 64 * GiB, dtype=np.int64), rng_w.integers(1, 9, N)], axis=1)], axis=1)
    ptr, sel = _csr_from_mask(mask)
    nodes = Table(hdr, "spx_node_objects", n_nodes=N, alloc_cpu_milli=nodes.array("alloc_cpu_milli"),
                  alloc_mem=nodes.array("alloc_mem"), alloc_eph=nodes.array("alloc_eph"), alloc_pods=nodes.array("alloc_pods"),
                  scalar_ptr=ptr, scalar_res=res.reshape(-1)[sel], scalar_qty=qty.reshape(-1)[sel],
                  cap_cpu_milli=nodes.array("cap_cpu_milli"), region=nodes.array("region"), zone=nodes.array("zone"))
    return {
        "nodes": nodes,
        "pods": synth_pods(hdr, n_pods, seed, device_res=RES_DEVICE, hugepage_res=RES_HUGEPAGES_2MI,
                           device2_res=RES_DEVICE2 if wide else -1, hugepage2_res=RES_HUGEPAGES_1GI if wide else -1),
        "nrt": synth_nrt(hdr, nodes, seed, vary=vary, wide=wide),
        "rc": nrt_resource_classes(hdr, wide),
    }


# ------------------------------------------------------------------ network-aware (config #4)
def synth_network(hdr: Header, nodes: Table, n_groups: int, seed: int = SEED, n_regions: int = 8, zones_per_region: int = 8,
                  placed_per_group: int = 10):
    """AppGroup + NetworkTopology objects: AppGroups of 11 workloads shaped like onlineboutique
    (networkoverhead_test.go:226-307) with D in [0,7] dependencies each and MaxNetworkCost in {5,10,20,50,100};
    10 already-placed pods per group on random nodes (as the reference benchmarks, :359-370); a 3-tier topology:
    dense zone->zone costs in [1,50] inside a region with 5% of entries missing, region->region in [20,100]."""
    rng = np.random.default_rng(seed + 7)
    N = nodes.struct.n_nodes
    W = 11
    G = n_groups
    n_wl = G * W
    wl_ptr = np.arange(G + 1, dtype=np.int32) * W
    wl_selector = np.tile(np.arange(W, dtype=np.int32), G)
    nd = rng.integers(0, 8, n_wl)
    nd = np.where(rng.random(n_wl) < 0.5, 0, nd)  # most workloads of onlineboutique have no dependencies
    dep_ptr = np.zeros(n_wl + 1, dtype=np.int32)
    np.cumsum(nd, out=dep_ptr[1:])
    n_dep = int(dep_ptr[-1])
    dep_selector = rng.integers(0, W, n_dep).astype(np.int32)
    dep_max = rng.choice(np.array([5, 10, 20, 50, 100], dtype=np.int64), n_dep)
    topo_ptr = wl_ptr.copy()
    topo_selector = wl_selector.copy()  # sorted by selector, as the controller writes it
    topo_index = (np.argsort(rng.random((G, W)), axis=1) + 1).astype(np.int32).reshape(-1)
    S = placed_per_group
    placed_ptr = np.arange(G + 1, dtype=np.int32) * S
    placed_selector = rng.integers(0, W, G * S).astype(np.int32)
    placed_node = rng.integers(0, N, G * S).astype(np.int32)
    ag = Table(hdr, "spx_appgroup_objects", n_groups=G, wl_ptr=wl_ptr, wl_selector=wl_selector, dep_ptr=dep_ptr,
               dep_selector=dep_selector, dep_max_cost=dep_max, topo_ptr=topo_ptr, topo_selector=topo_selector,
               topo_index=topo_index, placed_ptr=placed_ptr, placed_selector=placed_selector, placed_node=placed_node)
    Rg, Zc = n_regions, n_regions * zones_per_region
    rmask = ~np.eye(Rg, dtype=bool)
    rc_ptr, rsel = _csr_from_mask(rmask)
    rdest = np.tile(np.arange(Rg, dtype=np.int32), (Rg, 1))
    rcost = rng.integers(20, 101, (Rg, Rg))
    same_region = (np.arange(Zc)[:, None] // zones_per_region) == (np.arange(Zc)[None, :] // zones_per_region)
    zmask = same_region & ~np.eye(Zc, dtype=bool) & (rng.random((Zc, Zc)) >= 0.05)
    zc_ptr, zsel = _csr_from_mask(zmask)
    zdest = np.tile(np.arange(Zc, dtype=np.int32), (Zc, 1))
    zcost = rng.integers(1, 51, (Zc, Zc))
    nt = Table(hdr, "spx_nettopo_objects", n_regions=Rg, n_zones=Zc, rc_ptr=rc_ptr, rc_dest=rdest.reshape(-1)[rsel],
               rc_cost=rcost.reshape(-1)[rsel].astype(np.int64), zc_ptr=zc_ptr, zc_dest=zdest.reshape(-1)[zsel],
               zc_cost=zcost.reshape(-1)[zsel].astype(np.int64))
    return ag, nt


def network_snapshot(hdr: Header, n_nodes: int, n_pods: int, seed: int = SEED, pods_per_group: int = 100) -> Dict[str, Table]:
    """Object tables for BASELINE.json config #4 (NetworkOverhead + TopologicalSort)."""
    nodes = synth_nodes(hdr, n_nodes, seed)
    n_groups = max(1, n_pods // pods_per_group)
    ag, nt = synth_network(hdr, nodes, n_groups, seed)
    return {"nodes": nodes, "pods": synth_pods(hdr, n_pods, seed, n_appgroups=n_groups), "appgroups": ag, "nettopo": nt,
            "rc": resource_classes(hdr)}


# ------------------------------------------------------------------ CapacityScheduling (config #5's PreFilter gate)
def synth_quota(hdr: Header, pods: Table, seed: int = SEED, n_namespaces: int = 100, n_nominated: int = 300, device_res: int = -1,
                hugepage_res: int = -1, sized_for_batch: bool = False) -> Table:
    """ElasticQuotas for Q namespaces (Q=100): 85% of namespaces carry a quota; Used is drawn
    around Min so that both PreFilter gates fire for a visible share of pods; nominated pods with priorities in
    {0,100,1000}, a few of them being pending pods themselves (the uid exclusion, capacity_scheduling.go:239)."""
    rng = np.random.default_rng(seed + 8)
    NS, S = n_namespaces, 8
    has = rng.random(NS) < 0.85
    mn = np.zeros((NS, S), dtype=np.int64)
    mx = np.zeros((NS, S), dtype=np.int64)
    us = np.zeros((NS, S), dtype=np.int64)
    mn[:, 0] = rng.integers(50, 400, NS) * 1000
    mn[:, 1] = rng.integers(100, 2000, NS) * GiB
    mn[:, 2] = rng.integers(0, 100, NS) * GiB
    mx[:, :3] = (mn[:, :3] * rng.uniform(1.0, 2.0, (NS, 3))).astype(np.int64)
    no_max = rng.random(NS) < 0.1
    mx[no_max, :3] = (1 << 63) - 1
    us[:, :3] = (mn[:, :3] * rng.uniform(0.2, 1.3, (NS, 3))).astype(np.int64)
    if sized_for_batch:
        # Quotas in proportion to what the batch's namespaces ask for (the fixed sizes above suit the frozen sweep's PreFilter
        # hit rate; scheduled one after the other against them, 89 % of a 62.5k-pod batch ends over Max).  Min = 0.7 - 1.4 x the
        # namespace's total request, Max = 1 - 2 x Min, Used = 0 - 0.25 x Min.
        P = pods.struct.n_pods
        cp, qp = pods.array("ctr_ptr"), pods.array("req_ptr")
        ctr_of = np.repeat(np.arange(len(qp) - 1), np.diff(qp))
        pod_of_ctr = np.repeat(np.arange(P), np.diff(cp))
        pod_of_req = pod_of_ctr[ctr_of]
        ns_of_req = pods.array("ns")[pod_of_req]
        rr, qq = pods.array("req_res"), pods.array("req_qty")
        for slot, res in ((0, 0), (1, 1)):
            demand = np.bincount(ns_of_req[rr == res], weights=qq[rr == res].astype(np.float64), minlength=NS)
            mn[:, slot] = (np.maximum(demand, 1.0) * rng.uniform(0.7, 1.4, NS)).astype(np.int64)
        mx[:, :2] = (mn[:, :2] * rng.uniform(1.0, 2.0, (NS, 2))).astype(np.int64)
        mx[no_max, :2] = (1 << 63) - 1
        us[:, :2] = (mn[:, :2] * rng.uniform(0.0, 0.25, (NS, 2))).astype(np.int64)
    present = np.zeros(NS, dtype=np.uint8)
    n_scalar = 0
    scalar_res = np.zeros(4, dtype=np.int32)
    if device_res >= 0:
        n_scalar = 1
        scalar_res[0] = device_res
        mn[:, 4] = rng.integers(0, 64, NS)
        mx[:, 4] = mn[:, 4] + rng.integers(0, 64, NS)
        us[:, 4] = rng.integers(0, 80, NS)
        if sized_for_batch:  # ~10 % of the pods ask for 1-2 devices
            per_ns = max(1, pods.struct.n_pods // NS)
            mn[:, 4] = rng.integers(per_ns // 8, per_ns // 3 + 2, NS)
            mx[:, 4] = mn[:, 4] + rng.integers(0, per_ns // 4 + 2, NS)
            us[:, 4] = rng.integers(0, per_ns // 40 + 2, NS)
        present[:] = 1 << 4
    hp_slot = -1
    if hugepage_res >= 0:  # pods may request hugepages: a scalar resource the quotas do not bound (no key in Min/Max)
        scalar_res[n_scalar] = hugepage_res
        hp_slot = 4 + n_scalar
        n_scalar += 1
    min_present = np.where(rng.random(NS) < 0.9, present, 0).astype(np.uint8)  # some quotas do not list the device in Min
    if sized_for_batch:
        # every quota lists every scalar the batch asks for: a scalar missing from Min counts as Min = 0 in the AGGREGATE gate
        # (cmp2's LowerBoundOfMin, elasticquota.go:193-221 via aggregatedUsedOverMinWith :48-59), so one nominated pod with
        # hugepages anywhere in the cluster turns every quota'd pod away — faithful, and a degenerate queue to time a commit loop on
        min_present = present.copy()
        if hp_slot >= 0:
            mn[:, hp_slot] = 1 << 44
            mx[:, hp_slot] = 1 << 45
            present = (present | (1 << hp_slot)).astype(np.uint8)
            min_present = present.copy()
    P = pods.struct.n_pods
    nom_ns = rng.integers(0, NS, n_nominated).astype(np.int32)
    nom_prio = rng.choice(np.array([0, 100, 1000], dtype=np.int32), n_nominated)
    nom_pods = synth_pods(hdr, max(n_nominated, 1), seed=seed + 99, device_res=device_res, hugepage_res=hugepage_res, n_namespaces=NS)
    pend = np.full(n_nominated, -1, dtype=np.int64)
    k = min(n_nominated // 10, P)
    if k:
        idx = rng.choice(P, k, replace=False)
        pend[:k] = idx
        nom_ns[:k] = pods.array("ns")[idx]  # the same pod lives in the same namespace
    return Table(hdr, "spx_quota_objects", n_namespaces=NS, n_scalar_slots=n_scalar, scalar_res=scalar_res,
                 has_quota=has.astype(np.uint8), min=mn.reshape(-1), min_present=min_present, max=mx.reshape(-1),
                 max_present=present, used=us.reshape(-1), used_present=present, n_nominated=n_nominated, nom_ns=nom_ns,
                 nom_priority=nom_prio, nom_pending_index=pend, nom_pods=nom_pods)


# ------------------------------------------------------------------ full profile (config #5)
def full_snapshot(hdr: Header, n_nodes: int, n_pods: int, seed: int = SEED, pods_per_group: int = 100,
                  n_namespaces: int = 100, quota_sized_for_batch: bool = False) -> Dict[str, Table]:
    """One snapshot carrying every table of the full profile: CapacityScheduling PreFilter + Allocatable + NRT +
    trimaran + network-aware (BASELINE.json config #5)."""
    snap = nrt_snapshot(hdr, n_nodes, n_pods, seed)  # nodes (with hugepages/devices), NRT, rc
    n_groups = max(1, n_pods // pods_per_group)
    snap["pods"] = synth_pods(hdr, n_pods, seed, device_res=RES_DEVICE, hugepage_res=RES_HUGEPAGES_2MI, n_appgroups=n_groups,
                              n_namespaces=n_namespaces)
    snap["metrics"] = synth_metrics(hdr, n_nodes, seed)
    snap["assigned"] = synth_assigned(hdr, n_nodes, seed)
    snap["appgroups"], snap["nettopo"] = synth_network(hdr, snap["nodes"], n_groups, seed)
    snap["quota"] = synth_quota(hdr, snap["pods"], seed, n_namespaces=n_namespaces, device_res=RES_DEVICE,
                                hugepage_res=RES_HUGEPAGES_2MI, sized_for_batch=quota_sized_for_batch)
    return snap
