"""Builders for the "object tables" of include/spx.h from plain Python descriptions.

This is the Python stand-in for what the Go shim does when it marshals *v1.Pod / NodeInfo /
watcher.WatcherMetrics into flat C arrays.  Tests use it to write cases the
way the reference's table-driven tests write them (st.MakePod()..., makeNodeInfo(...)).
"""
from __future__ import annotations

from fractions import Fraction
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

from ._abi import Header, Table

RES_CPU, RES_MEMORY, RES_EPHEMERAL, RES_PODS, RES_STORAGE = 0, 1, 2, 3, 4
RES_FIRST_DYNAMIC = 8
RC_HUGEPAGE, RC_NATIVE, RC_SCALAR = 1, 2, 4

_SUFFIX = {
    "n": Fraction(1, 10**9), "u": Fraction(1, 10**6), "m": Fraction(1, 1000), "": Fraction(1),
    "k": Fraction(10**3), "M": Fraction(10**6), "G": Fraction(10**9), "T": Fraction(10**12),
    "P": Fraction(10**15), "E": Fraction(10**18),
    "Ki": Fraction(2**10), "Mi": Fraction(2**20), "Gi": Fraction(2**30), "Ti": Fraction(2**40),
    "Pi": Fraction(2**50), "Ei": Fraction(2**60),
}


def parse_quantity(q) -> Fraction:
    """resource.MustParse: exact decimal value of a Kubernetes quantity string (or number)."""
    if isinstance(q, (int, np.integer)):
        return Fraction(int(q))
    if isinstance(q, Fraction):
        return q
    s = str(q).strip()
    for suf in sorted(_SUFFIX, key=len, reverse=True):
        if suf and s.endswith(suf):
            return Fraction(s[: -len(suf)]) * _SUFFIX[suf]
    if "e" in s or "E" in s:
        mant, exp = s.lower().split("e")
        return Fraction(mant) * Fraction(10) ** int(exp)
    return Fraction(s)


def _ceil(fr: Fraction) -> int:
    """Quantity.Value() / MilliValue() rounding: inexact values go AWAY from zero for either sign (apimachinery
    resource/math.go negativeScaleInt64: value++ for positive, value-- for negative amounts) — a ceiling only for v >= 0"""
    if fr < 0:
        return -_ceil(-fr)
    return -((-fr.numerator) // fr.denominator)


def is_native_resource(name: str) -> bool:  # v1helper.IsNativeResource
    return "/" not in name or "kubernetes.io/" in name


def is_hugepage(name: str) -> bool:  # v1helper.IsHugePageResourceName
    return name.startswith("hugepages-")


def is_scalar_resource_name(name: str) -> bool:  # schedutil.IsScalarResourceName
    extended = (not is_native_resource(name)) and not name.startswith("requests.")
    prefixed_native = "kubernetes.io/" in name
    attachable = name.startswith("attachable-volumes-")
    return extended or is_hugepage(name) or prefixed_native or attachable


class Resources:
    """Interns resource names to the canonical int ids of spx.h and keeps their class flags."""

    FIXED = {"cpu": RES_CPU, "memory": RES_MEMORY, "ephemeral-storage": RES_EPHEMERAL, "pods": RES_PODS,
             "storage": RES_STORAGE}

    def __init__(self):
        self.ids: Dict[str, int] = dict(self.FIXED)
        self.names: Dict[int, str] = {v: k for k, v in self.FIXED.items()}
        self._next = RES_FIRST_DYNAMIC

    def id(self, name: str) -> int:
        if name not in self.ids:
            self.ids[name] = self._next
            self.names[self._next] = name
            self._next += 1
        return self.ids[name]

    def canonical(self, name: str, q) -> int:
        """cpu -> MilliValue(), everything else -> Value(); both round inexact values away from zero."""
        fr = parse_quantity(q)
        return _ceil(fr * 1000) if name == "cpu" else _ceil(fr)

    def flags(self) -> np.ndarray:
        out = np.zeros(max(self._next, RES_FIRST_DYNAMIC), dtype=np.uint8)
        for name, i in self.ids.items():
            f = 0
            if is_hugepage(name):
                f |= RC_HUGEPAGE
            if is_native_resource(name):
                f |= RC_NATIVE
            if is_scalar_resource_name(name):
                f |= RC_SCALAR
            out[i] = f
        return out

    def table(self, hdr: Header) -> Table:
        fl = self.flags()
        return Table(hdr, "spx_resource_classes", n_res=len(fl), flags=fl)


def _csr(lists: Sequence[Sequence]) -> np.ndarray:
    ptr = np.zeros(len(lists) + 1, dtype=np.int32)
    for i, l in enumerate(lists):
        ptr[i + 1] = ptr[i] + len(l)
    return ptr


def _rl(res: Resources, rl: Optional[dict]):
    """v1.ResourceList -> [(id, canonical qty)] keeping key presence (zero quantities stay)."""
    if not rl:
        return []
    return [(res.id(k), res.canonical(k, v)) for k, v in rl.items()]


def container(requests: Optional[dict] = None, limits: Optional[dict] = None, sidecar: bool = False) -> dict:
    return {"requests": requests or {}, "limits": limits or {}, "sidecar": sidecar}


def pod(containers: Iterable[dict] = (), init_containers: Iterable[dict] = (), overhead: Optional[dict] = None,
        priority: int = 0, queue_ts: int = 0, appgroup: int = -1, selector: int = -1, ns: int = 0) -> dict:
    return {"containers": list(containers), "init_containers": list(init_containers), "overhead": overhead,
            "priority": priority, "queue_ts": queue_ts, "appgroup": appgroup, "selector": selector, "ns": ns}


def build_pod_objects(hdr: Header, res: Resources, pods: Sequence[dict]) -> Table:
    kinds, reqs, lims, ovhs, per_pod = [], [], [], [], []
    for p in pods:
        ctrs = [(c, 2 if c.get("sidecar") else 1) for c in p.get("init_containers", [])]
        ctrs += [(c, 0) for c in p.get("containers", [])]
        per_pod.append(ctrs)
        for c, k in ctrs:
            kinds.append(k)
            reqs.append(_rl(res, c.get("requests")))
            lims.append(_rl(res, c.get("limits")))
        ovhs.append(_rl(res, p.get("overhead")))
    flat = lambda ls, j: [x[j] for l in ls for x in l]
    return Table(
        hdr, "spx_pod_objects",
        n_pods=len(pods),
        ctr_ptr=_csr(per_pod), ctr_kind=np.array(kinds, dtype=np.uint8),
        req_ptr=_csr(reqs), req_res=flat(reqs, 0), req_qty=flat(reqs, 1),
        lim_ptr=_csr(lims), lim_res=flat(lims, 0), lim_qty=flat(lims, 1),
        ovh_ptr=_csr(ovhs), ovh_res=flat(ovhs, 0), ovh_qty=flat(ovhs, 1),
        priority=[p.get("priority", 0) for p in pods],
        queue_ts=[p.get("queue_ts", 0) for p in pods],
        appgroup=[p.get("appgroup", -1) for p in pods],
        selector=[p.get("selector", -1) for p in pods],
        ns=[p.get("ns", 0) for p in pods],
    )


def pod_rows(hdr: Header, pods: Table, begin: int, end: int) -> Table:
    """rows [begin, end) of a pod object table as a table of its own, without copying: the per-pod columns are offset, the
    per-container columns shared — their CSR offsets (ctr_ptr / req_ptr / lim_ptr / ovh_ptr values) are absolute.  What a rank
    of a sharded batch hands to the flatteners (MultiEngine)."""
    cols = {k: pods.array(k) for k in pods._keep}
    for k in ("priority", "queue_ts", "appgroup", "selector", "ns"):
        if k in cols:
            cols[k] = cols[k][begin:end]
    for k in ("ctr_ptr", "ovh_ptr"):
        if k in cols:
            cols[k] = cols[k][begin:end + 1]
    return Table(hdr, "spx_pod_objects", n_pods=end - begin, **cols)


def node(allocatable: Optional[dict] = None, capacity: Optional[dict] = None, region: int = -1, zone: int = -1) -> dict:
    """allocatable/capacity are v1.ResourceList-like dicts; capacity defaults to allocatable
    (st.MakeNode().Capacity(...) sets both in the reference's tests)."""
    allocatable = allocatable or {}
    return {"allocatable": allocatable, "capacity": capacity if capacity is not None else allocatable,
            "region": region, "zone": zone}


def build_node_objects(hdr: Header, res: Resources, nodes: Sequence[dict]) -> Table:
    def get(n, key, name):
        rl = n[key]
        return res.canonical(name, rl[name]) if name in rl else 0

    scalars = []
    for n in nodes:
        sc = []
        for k, v in n["allocatable"].items():
            if k in ("cpu", "memory", "ephemeral-storage", "pods"):
                continue
            if is_scalar_resource_name(k):  # framework.Resource.Add keeps only scalar names
                sc.append((res.id(k), res.canonical(k, v)))
        scalars.append(sc)
    return Table(
        hdr, "spx_node_objects",
        n_nodes=len(nodes),
        alloc_cpu_milli=[get(n, "allocatable", "cpu") for n in nodes],
        alloc_mem=[get(n, "allocatable", "memory") for n in nodes],
        alloc_eph=[get(n, "allocatable", "ephemeral-storage") for n in nodes],
        alloc_pods=[get(n, "allocatable", "pods") for n in nodes],
        scalar_ptr=_csr(scalars),
        scalar_res=[x[0] for l in scalars for x in l],
        scalar_qty=[x[1] for l in scalars for x in l],
        cap_cpu_milli=[get(n, "capacity", "cpu") for n in nodes],
        region=[n.get("region", -1) for n in nodes],
        zone=[n.get("zone", -1) for n in nodes],
    )


MT = {"CPU": 0, "Memory": 1}
MO = {"AVG": 0, "STD": 1, "Latest": 2, "": 3}


def build_metrics_objects(hdr: Header, n_nodes: int, node_metrics: Optional[Dict[int, Optional[list]]],
                          window_end: int = 0) -> Table:
    """node_metrics: None = NodeMetricsMap nil ("404 resp from watcher"); else {node index: [(type, op, value)...]};
    a node missing from the dict has no entry in the map; a value of None is a nil Metrics slice."""
    present = np.zeros(n_nodes, dtype=np.uint8)
    isnil = np.zeros(n_nodes, dtype=np.uint8)
    lists = [[] for _ in range(n_nodes)]
    if node_metrics is not None:
        for i, ms in node_metrics.items():
            present[i] = 1
            if ms is None:
                isnil[i] = 1
            else:
                lists[i] = [(MT.get(t, 2), MO.get(o, 4), float(v)) for t, o, v in ms]
    return Table(
        hdr, "spx_metrics_objects",
        map_is_nil=1 if node_metrics is None else 0,
        window_end=window_end,
        node_present=present, node_metrics_nil=isnil,
        m_ptr=_csr(lists),
        m_type=np.array([x[0] for l in lists for x in l], dtype=np.uint8),
        m_op=np.array([x[1] for l in lists for x in l], dtype=np.uint8),
        m_value=np.array([x[2] for l in lists for x in l], dtype=np.float64),
    )


def build_assigned_objects(hdr: Header, res: Resources, n_nodes: int, entries: Dict[int, list]) -> Table:
    """entries: {node index: [(timestamp_unix, pod dict), ...]} — ScheduledPodsCache image."""
    pods, ts, per_node = [], [], [[] for _ in range(n_nodes)]
    for ni in range(n_nodes):  # CSR order: entry index == index into `pods`
        for t, p in entries.get(ni, []):
            per_node[ni].append(len(pods))
            pods.append(p)
            ts.append(t)
    ptable = build_pod_objects(hdr, res, pods)
    return Table(hdr, "spx_assigned_objects", e_ptr=_csr(per_node), e_ts_unix=np.array(ts, dtype=np.int64),
                 e_pod=np.array([i for l in per_node for i in l], dtype=np.int32), pods=ptable)


def build_node_pods_objects(hdr: Header, res: Resources, n_nodes: int, pods_on_node: Dict[int, list]) -> Table:
    """pods_on_node: {node index: [pod dict, ...]} — framework.NodeInfo.GetPods() image."""
    pods, per_node = [], [[] for _ in range(n_nodes)]
    for ni in range(n_nodes):
        for p in pods_on_node.get(ni, []):
            per_node[ni].append(len(pods))
            pods.append(p)
    ptable = build_pod_objects(hdr, res, pods)
    return Table(hdr, "spx_node_pods_objects", p_ptr=_csr(per_node),
                 p_pod=np.array([i for l in per_node for i in l], dtype=np.int32), pods=ptable)


# ------------------------------------------------------------------ NodeResourceTopology
LEGACY_POLICIES = {  # topologyv1alpha2.TopologyManagerPolicy -> (policy << 1) | scope   nodeconfig/topologymanager.go:141-160
    "SingleNUMANodeContainerLevel": (3 << 1) | 0, "SingleNUMANodePodLevel": (3 << 1) | 1,
    "BestEffortContainerLevel": (1 << 1) | 0, "BestEffortPodLevel": (1 << 1) | 1,
    "RestrictedContainerLevel": (2 << 1) | 0, "RestrictedPodLevel": (2 << 1) | 1,
}
TM_POLICY = {"none": 0, "best-effort": 1, "restricted": 2, "single-numa-node": 3}
TM_SCOPE = {"container": 0, "pod": 1}


def numa_name_to_id(name: str) -> int:  # numanode.NameToID: "node-<id>"
    if not name.startswith("node-"):
        return -1
    try:
        return int(name[len("node-"):])
    except ValueError:
        return -1


def nrt(zones: Sequence[dict], policies: Sequence[str] = (), attributes: Optional[dict] = None) -> dict:
    """zones: [{"name": "node-0", "type": "Node", "resources": {name: available} | [(name, capacity, available)] |
    [(name, capacity, allocatable, available)], "costs": {"node-1": 12, ...}}]"""
    return {"zones": list(zones), "policies": list(policies), "attributes": dict(attributes or {})}


def build_nrt_objects(hdr: Header, res: Resources, nrts: Sequence[Optional[dict]], fresh: Optional[Sequence[bool]] = None,
                      assumed: Optional[Dict[int, list]] = None) -> Table:
    n = len(nrts)
    legacy, a_scope, a_policy, a_max = [], [], [], []
    zone_lists, zres, zcost, z_is_node, z_id, zalloc = [], [], [], [], [], []
    for t in nrts:
        lp, sc, po, mx, zs = -1, -1, -1, -1, []
        if t is not None:
            pol = t.get("policies") or []
            if pol:
                lp = LEGACY_POLICIES.get(pol[0], -1)
            for k, v in (t.get("attributes") or {}).items():
                if k == "topologyManagerScope" and v in TM_SCOPE:
                    sc = TM_SCOPE[v]
                elif k == "topologyManagerPolicy" and v in TM_POLICY:
                    po = TM_POLICY[v]
                elif k == "topologyManagerMaxNUMANodes":
                    try:
                        mx = int(v) if int(v) > 1 else -1
                    except ValueError:
                        mx = -1
            for z in t["zones"]:
                zs.append(z)
                z_is_node.append(1 if z.get("type", "Node") == "Node" else 0)
                z_id.append(numa_name_to_id(z["name"]))
                rl = z.get("resources") or {}
                if isinstance(rl, dict):
                    zres.append([(res.id(k), res.canonical(k, v)) for k, v in rl.items()])
                    zalloc.extend(res.canonical(k, v) for k, v in rl.items())
                else:
                    zres.append([(res.id(t4[0]), res.canonical(t4[0], t4[-1])) for t4 in rl])
                    zalloc.extend(res.canonical(t4[0], t4[2] if len(t4) == 4 else t4[1]) for t4 in rl)  # allocatable, else capacity
                costs = z.get("costs") or {}
                items = costs.items() if isinstance(costs, dict) else costs
                zcost.append([(numa_name_to_id(k), int(v)) for k, v in items])
        legacy.append(lp), a_scope.append(sc), a_policy.append(po), a_max.append(mx)
        zone_lists.append(zs)
    assumed = assumed or {}
    a_lists, per_node = [], []
    for i in range(n):
        lists = [_rl(res, rl) for rl in assumed.get(i, [])]
        per_node.append(lists)
        a_lists.extend(lists)
    return Table(
        hdr, "spx_nrt_objects", n_nodes=n,
        has_nrt=np.array([t is not None for t in nrts], dtype=np.uint8),
        fresh=np.ones(n, dtype=np.uint8) if fresh is None else np.array(fresh, dtype=np.uint8),
        legacy_policy=np.array(legacy, dtype=np.int8), attr_scope=np.array(a_scope, dtype=np.int8),
        attr_policy=np.array(a_policy, dtype=np.int8), attr_max_numa=np.array(a_max, dtype=np.int32),
        zone_ptr=_csr(zone_lists), zone_is_node=np.array(z_is_node, dtype=np.uint8), zone_numa_id=np.array(z_id, dtype=np.int32),
        zres_ptr=_csr(zres), zres_res=[x[0] for l in zres for x in l], zres_avail=[x[1] for l in zres for x in l],
        zcost_ptr=_csr(zcost), zcost_numa_id=[x[0] for l in zcost for x in l], zcost_value=[x[1] for l in zcost for x in l],
        assumed_ptr=_csr(per_node), arl_ptr=_csr(a_lists), arl_res=[x[0] for l in a_lists for x in l],
        arl_qty=[x[1] for l in a_lists for x in l],
        zres_allocatable=np.array(zalloc, dtype=np.int64),
    )


def nrt_params(hdr: Header, res: Resources, strategy: str = "LeastAllocated", weights: Optional[dict] = None) -> Table:
    strat = {"MostAllocated": 0, "BalancedAllocation": 1, "LeastAllocated": 2, "LeastNUMANodes": 3}[strategy]
    weights = weights or {}
    return Table(hdr, "spx_nrt_params", strategy=strat, n_weights=len(weights),
                 weight_res=np.array([res.id(k) for k in weights], dtype=np.int32),
                 weight=np.array(list(weights.values()), dtype=np.int64))


def node_from_zones(zones: Sequence[dict], extra: Optional[dict] = None) -> dict:
    """makeResourceListFromZones (objects.go:83-95): node allocatable = Σ zone Available per resource (+extras)."""
    total: Dict[str, Fraction] = {}
    for z in zones:
        rl = z.get("resources") or {}
        items = rl.items() if isinstance(rl, dict) else [(t[0], t[-1]) for t in rl]
        for k, v in items:
            total[k] = total.get(k, Fraction(0)) + parse_quantity(v)
    for k, v in (extra or {}).items():
        total[k] = parse_quantity(v)
    return node(total)


# ------------------------------------------------------------------ network-aware CRs
class Interner:
    """String -> dense id; `sorted_ids()` re-numbers so that id order == lexicographic order (needed for
    AppGroup workload selectors, which util.FindPodOrder compares as strings)."""

    def __init__(self, names: Iterable[str] = ()):
        self.ids: Dict[str, int] = {}
        for n in names:
            self.id(n)

    def id(self, name: Optional[str]) -> int:
        if name is None or name == "":
            return -1
        if name not in self.ids:
            self.ids[name] = len(self.ids)
        return self.ids[name]

    def freeze_sorted(self) -> None:
        self.ids = {n: i for i, n in enumerate(sorted(self.ids))}

    def __len__(self):
        return len(self.ids)


def build_appgroup_objects(hdr: Header, selectors: Interner, groups: Sequence[dict], node_index: Dict[str, int]) -> Table:
    """groups: [{"workloads": [{"selector": s, "dependencies": [(selector, max_network_cost), ...]}],
    "topology_order": [(selector, index)], "placed": [(selector, hostname)]}]; selectors must be frozen-sorted."""
    wl, deps, topo, placed = [], [], [], []
    for g in groups:
        ws = g.get("workloads", [])
        wl.append([selectors.ids[w["selector"]] for w in ws])
        for w in ws:
            deps.append([(selectors.ids[s], int(c)) for s, c in w.get("dependencies", [])])
        topo.append([(selectors.ids[s], int(i)) for s, i in g.get("topology_order", [])])
        placed.append([(selectors.ids.get(s, -1), node_index.get(h, -1)) for s, h in g.get("placed", []) if h])
    return Table(
        hdr, "spx_appgroup_objects", n_groups=len(groups),
        wl_ptr=_csr(wl), wl_selector=[x for l in wl for x in l],
        dep_ptr=_csr(deps), dep_selector=[x[0] for l in deps for x in l], dep_max_cost=[x[1] for l in deps for x in l],
        topo_ptr=_csr(topo), topo_selector=[x[0] for l in topo for x in l], topo_index=[x[1] for l in topo for x in l],
        placed_ptr=_csr(placed), placed_selector=[x[0] for l in placed for x in l], placed_node=[x[1] for l in placed for x in l],
    )


def build_nettopo_objects(hdr: Header, regions: Interner, zones: Interner, region_costs: Dict[str, list], zone_costs: Dict[str, list]) -> Table:
    """region_costs / zone_costs: {origin: [(destination, cost), ...]} of the configured weights set."""
    for o, l in list(region_costs.items()):
        regions.id(o)
        for d, _ in l:
            regions.id(d)
    for o, l in list(zone_costs.items()):
        zones.id(o)
        for d, _ in l:
            zones.id(d)
    rc = [[] for _ in range(len(regions))]
    zc = [[] for _ in range(len(zones))]
    for o, l in region_costs.items():
        rc[regions.ids[o]] += [(regions.ids[d], int(c)) for d, c in l]
    for o, l in zone_costs.items():
        zc[zones.ids[o]] += [(zones.ids[d], int(c)) for d, c in l]
    return Table(hdr, "spx_nettopo_objects", n_regions=len(regions), n_zones=len(zones),
                 rc_ptr=_csr(rc), rc_dest=[x[0] for l in rc for x in l], rc_cost=[x[1] for l in rc for x in l],
                 zc_ptr=_csr(zc), zc_dest=[x[0] for l in zc for x in l], zc_cost=[x[1] for l in zc for x in l])


# ------------------------------------------------------------------ CapacityScheduling (ElasticQuota)
QUOTA_SLOTS = 8
UPPER_BOUND_OF_MAX = (1 << 63) - 1  # elasticquota.go:29


def _resource_vec(res: Resources, scalar_slots: List[int], rl):
    """framework.NewResource(rl) over the quota slot vector.  `rl` may be a ResourceList-like dict, or a dict
    with the framework.Resource field names the reference's tests use (MilliCPU, Memory, ..., ScalarResources)."""
    v = [0] * QUOTA_SLOTS
    present = 0
    if rl is None:
        return v, present
    if any(k in rl for k in ("MilliCPU", "Memory", "EphemeralStorage", "AllowedPodNumber", "ScalarResources")):
        v[0], v[1], v[2], v[3] = (int(rl.get(k, 0)) for k in ("MilliCPU", "Memory", "EphemeralStorage", "AllowedPodNumber"))
        items = [(k, int(q)) for k, q in (rl.get("ScalarResources") or {}).items()]
    else:
        items = []
        for k, q in rl.items():
            if k == "cpu":
                v[0] += res.canonical(k, q)
            elif k == "memory":
                v[1] += res.canonical(k, q)
            elif k == "ephemeral-storage":
                v[2] += res.canonical(k, q)
            elif k == "pods":
                v[3] += res.canonical(k, q)
            elif is_scalar_resource_name(k):
                items.append((k, res.canonical(k, q)))
    for k, q in items:
        rid = res.id(k)
        if rid not in scalar_slots:
            scalar_slots.append(rid)
        s = 4 + scalar_slots.index(rid)
        if s >= QUOTA_SLOTS:
            raise ValueError("more scalar resources than quota slots")
        v[s] += q
        present |= 1 << s
    return v, present


def build_quota_objects(hdr: Header, res: Resources, namespaces: Sequence[Optional[dict]], nominated: Sequence[tuple] = ()) -> Table:
    """namespaces[i] = None (no ElasticQuota) or {"min": rl|None, "max": rl|None, "used": rl};
    nominated = [(namespace index, priority, pending pod index or -1, pod dict)].
    A nil Min/Max is replaced like newElasticQuotaInfo does (elasticquota.go:70-76)."""
    slots: List[int] = []
    has, mins, maxs, useds, mp, xp, up = [], [], [], [], [], [], []
    bound_min = {"cpu": "0", "memory": 0, "ephemeral-storage": 0}
    for ns in namespaces:
        has.append(1 if ns is not None else 0)
        ns = ns or {}
        mn, mnp = _resource_vec(res, slots, ns.get("min") if ns.get("min") is not None else bound_min)
        if ns.get("max") is not None:
            mx, mxp = _resource_vec(res, slots, ns["max"])
        else:
            mx, mxp = [UPPER_BOUND_OF_MAX, UPPER_BOUND_OF_MAX, UPPER_BOUND_OF_MAX, 0, 0, 0, 0, 0], 0
        us, usp = _resource_vec(res, slots, ns.get("used"))
        mins.append(mn), maxs.append(mx), useds.append(us), mp.append(mnp), xp.append(mxp), up.append(usp)
    nom_pods = build_pod_objects(hdr, res, [n[3] for n in nominated] or [pod()])
    for n in nominated:  # scalar requests of nominated pods need slots too
        for c in list(n[3].get("containers", [])) + list(n[3].get("init_containers", [])):
            _resource_vec(res, slots, c.get("requests"))
    return Table(
        hdr, "spx_quota_objects", n_namespaces=len(namespaces), n_scalar_slots=len(slots),
        scalar_res=np.array(slots + [0] * (4 - len(slots)), dtype=np.int32), has_quota=np.array(has, dtype=np.uint8),
        min=np.array(mins, dtype=np.int64).reshape(-1), min_present=np.array(mp, dtype=np.uint8),
        max=np.array(maxs, dtype=np.int64).reshape(-1), max_present=np.array(xp, dtype=np.uint8),
        used=np.array(useds, dtype=np.int64).reshape(-1), used_present=np.array(up, dtype=np.uint8),
        n_nominated=len(nominated), nom_ns=np.array([n[0] for n in nominated], dtype=np.int32),
        nom_priority=np.array([n[1] for n in nominated], dtype=np.int32),
        nom_pending_index=np.array([n[2] for n in nominated], dtype=np.int64), nom_pods=nom_pods,
    )
