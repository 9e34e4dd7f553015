"""Wire-format ingestion of the network-aware CRs: AppGroup and NetworkTopology JSON -> object tables,
against the independent Python builders, on the reference's example AppGroups (tests/golden/appgroup_manifests.json) and on the
fixtures of its network-aware unit tests (tests/golden/network.py) rendered in the CRD's shape; then the decoded tables drive
the oracle's NetworkOverhead Filter/Score to the reference's known answers.  CPU only."""
import json
from pathlib import Path

import numpy as np
import pytest

from golden import network as GN
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd.ingest import NrtIngest

GOLD = Path(__file__).parent / "golden"
REGION, ZONE = "topology.kubernetes.io/region", "topology.kubernetes.io/zone"


def col(struct, name, n):
    return np.ctypeslib.as_array(getattr(struct, name), (n,)).tolist() if n else []


def appgroup_cr(name, g):
    wl = lambda s: {"kind": "Deployment", "name": s, "selector": s, "apiVersion": "apps/v1", "namespace": "default"}
    return {"apiVersion": "appgroup.diktyo.x-k8s.io/v1alpha1", "kind": "AppGroup", "metadata": {"name": name},
            "spec": {"numMembers": len(g["workloads"]), "topologySortingAlgorithm": "KahnSort",
                     "workloads": [{"workload": wl(w["selector"]), "dependencies": [{"workload": wl(s), "maxNetworkCost": c} for s, c in w["dependencies"]]}
                                   for w in g["workloads"]]},
            "status": {"runningWorkloads": 0, "topologyOrder": [{"workload": wl(s), "index": i} for s, i in g["topology_order"]]}}


def nettopo_cr(region_costs, zone_costs, name="UserDefined"):
    def tl(key, costs):
        return {"topologyKey": key, "originList": [{"origin": o, "costList": [{"destination": d, "bandwidthCapacity": "1Gi", "networkCost": c} for d, c in l]}
                                                   for o, l in costs.items()]}
    return {"apiVersion": "networktopology.diktyo.x-k8s.io/v1alpha1", "kind": "NetworkTopology", "metadata": {"name": "nt", "namespace": "default"},
            "spec": {"configmapName": "netperfMetrics",
                     "weights": [{"name": "Other", "topologyList": [tl(REGION, {"x": [("y", 99)]})]},
                                 {"name": name, "topologyList": [tl(REGION, region_costs), tl(ZONE, zone_costs)]}]}}


def groups_equal(a, b):
    n = a.n_groups
    assert n == b.n_groups
    nw = a.wl_ptr[n]
    assert col(a, "wl_ptr", n + 1) == col(b, "wl_ptr", n + 1) and col(a, "wl_selector", nw) == col(b, "wl_selector", nw)
    nd = a.dep_ptr[nw]
    assert col(a, "dep_ptr", nw + 1) == col(b, "dep_ptr", nw + 1)
    assert col(a, "dep_selector", nd) == col(b, "dep_selector", nd) and col(a, "dep_max_cost", nd) == col(b, "dep_max_cost", nd)
    nt = a.topo_ptr[n]
    assert col(a, "topo_ptr", n + 1) == col(b, "topo_ptr", n + 1)
    assert col(a, "topo_selector", nt) == col(b, "topo_selector", nt) and col(a, "topo_index", nt) == col(b, "topo_index", nt)
    assert col(a, "placed_ptr", n + 1) == [0] * (n + 1)


def sorted_selectors(groups):
    sel = O.Interner()
    for g in groups:
        for w in g["workloads"]:
            sel.id(w["selector"])
            for s, _ in w["dependencies"]:
                sel.id(s)
        for s, _ in g["topology_order"]:
            sel.id(s)
    sel.freeze_sorted()
    return sel


def test_reference_example_appgroups(hdr):
    docs = json.loads((GOLD / "appgroup_manifests.json").read_text())
    assert [d["metadata"]["name"] for d in docs] == ["a1", "redis-cluster"]
    groups = []
    for d in docs:
        groups.append({"workloads": [{"selector": w["workload"]["selector"],
                                      "dependencies": [(x["workload"]["selector"], x.get("maxNetworkCost", 0)) for x in w.get("dependencies", [])]}
                                     for w in d["spec"]["workloads"]],
                       "topology_order": [(t["workload"]["selector"], t["index"]) for t in (d.get("status") or {}).get("topologyOrder", [])]})
    sel = sorted_selectors(groups)
    want = O.build_appgroup_objects(hdr, sel, groups, {})
    with NrtIngest(["n0"]) as ing:
        assert ing.feed_appgroups(json.dumps({"items": docs}).encode()) == 2
        groups_equal(ing.appgroup_objects().struct, want.struct)
        assert ing.name_id("appgroup", "redis-cluster") == 1
        for s, i in sel.ids.items():
            assert ing.name_id("selector", s) == i     # lexicographic ids: "P1" < "P2" < "P3" < ...
        t = ing.appgroup_objects().struct
        assert col(t, "dep_max_cost", t.dep_ptr[t.wl_ptr[1]])[:2] == [30, 20]   # appGroup-example.yaml: P1 -> P2 (30), P2 -> P3 (20)


def test_unit_test_fixtures_round_trip(hdr):
    groups = [GN.APPGROUP_BASIC, GN.ONLINEBOUTIQUE]
    sel = sorted_selectors(groups)
    want = O.build_appgroup_objects(hdr, sel, groups, {})
    regions, zones = O.Interner(), O.Interner()
    want_nt = O.build_nettopo_objects(hdr, regions, zones, GN.REGION_COSTS, GN.ZONE_COSTS)
    with NrtIngest(["n0"]) as ing:
        ing.feed_appgroups(json.dumps([appgroup_cr("basic", groups[0]), appgroup_cr("onlineboutique", groups[1])]).encode())
        groups_equal(ing.appgroup_objects().struct, want.struct)
        ing.feed_nettopo(json.dumps(nettopo_cr(GN.REGION_COSTS, GN.ZONE_COSTS)).encode(), "UserDefined")
        a, b = ing.nettopo_objects().struct, want_nt.struct
        assert (a.n_regions, a.n_zones) == (b.n_regions, b.n_zones) == (2, 4)
        for ptr, dest, cost, n in (("rc_ptr", "rc_dest", "rc_cost", 2), ("zc_ptr", "zc_dest", "zc_cost", 4)):
            m = getattr(a, ptr)[n]
            assert col(a, ptr, n + 1) == col(b, ptr, n + 1) and col(a, dest, m) == col(b, dest, m) and col(a, cost, m) == col(b, cost, m)
        assert ing.name_id("region", "us-west-1") == regions.ids["us-west-1"] and ing.name_id("zone", "Z3") == zones.ids["Z3"]
        # another weights set of the same CR
        ing.feed_nettopo(json.dumps(nettopo_cr(GN.REGION_COSTS, GN.ZONE_COSTS)).encode(), "Other")
        assert ing.nettopo_objects().struct.rc_ptr[ing.nettopo_objects().struct.n_regions] == 1



def _costs_of(t, which, names):
    """{origin: {destination: cost}} of the decoded region ('r') / zone ('z') table, names looked up by id"""
    ptr, dest, cost = (t.rc_ptr, t.rc_dest, t.rc_cost) if which == "r" else (t.zc_ptr, t.zc_dest, t.zc_cost)
    n = t.n_regions if which == "r" else t.n_zones
    out = {}
    for o in range(n):
        row = {names[dest[k]]: cost[k] for k in range(ptr[o], ptr[o + 1])}
        if row:
            out[names[o]] = row
    return out


def test_nettopo_lookups_follow_the_reference_binary_searches(hdr):
    """populateCostMap finds the topology key and the origin by BINARY SEARCH (util.FindTopologyKey / FindOriginCosts,
    util.go:156-191) over lists it sorts only for manually defined weights (networkoverhead.go:438-445, :462-465).  The decoder
    reproduces that, bug for bug: (a) the reference's own fixture lists the region key twice (networkoverhead_test.go:93,127) —
    the search lands on the first entry and zones are never found; (b) an unsorted NetperfCosts list (the controller writes it
    sorted; nothing re-sorts it) misses origins; (c) the same unsorted list under manual weights is sorted first and complete."""
    def tl(key, costs):
        return {"topologyKey": key, "originList": [{"origin": o, "costList": [{"destination": d, "networkCost": c} for d, c in l]} for o, l in costs]}

    def cr(name, topo):
        return json.dumps({"kind": "NetworkTopology", "spec": {"weights": [{"name": name, "topologyList": topo}]}}).encode()

    def names_of(ing, kind, candidates):
        out = {}
        for c in candidates:
            i = ing.name_id(kind, c)
            if i >= 0:
                out[i] = c
        return out

    r_costs = [("R1", [("R2", 50)]), ("R2", [("R1", 50)])]
    z_costs = [("Z1", [("Z2", 10)]), ("Z2", [("Z1", 10)])]
    with NrtIngest(["n0"]) as ing:
        # (a) both entries keyed "region": zones unreachable, regions = the FIRST entry
        ing.feed_nettopo(cr("UserDefined", [tl(REGION, r_costs), tl(REGION, z_costs)]), "UserDefined")
        t = ing.nettopo_objects().struct
        assert _costs_of(t, "r", names_of(ing, "region", ["R1", "R2", "Z1", "Z2"])) == {"R1": {"R2": 50}, "R2": {"R1": 50}}
        assert t.zc_ptr[t.n_zones] == 0
        # (b) NetperfCosts, origins out of order [C, A, B]: the search (mid = 1 -> "A") finds A and B, never C
        unsorted = [("C", [("A", 3)]), ("A", [("B", 1)]), ("B", [("A", 2)])]
        ing.feed_nettopo(cr("NetperfCosts", [tl(REGION, unsorted)]), "NetperfCosts")
        t = ing.nettopo_objects().struct
        assert _costs_of(t, "r", names_of(ing, "region", ["A", "B", "C"])) == {"A": {"B": 1}, "B": {"A": 2}}
        # keys out of order under NetperfCosts: [zone, region] — "region" < "zone": mid = 0 is "zone" > "region", high = -1: missed
        ing.feed_nettopo(cr("NetperfCosts", [tl(ZONE, z_costs), tl(REGION, r_costs)]), "NetperfCosts")
        t = ing.nettopo_objects().struct
        assert t.rc_ptr[t.n_regions] == 0 and _costs_of(t, "z", names_of(ing, "zone", ["Z1", "Z2"])) == {"Z1": {"Z2": 10}, "Z2": {"Z1": 10}}
        # (c) manual weights: sorted before the searches, nothing is missed
        ing.feed_nettopo(cr("UserDefined", [tl(ZONE, z_costs), tl(REGION, unsorted)]), "UserDefined")
        t = ing.nettopo_objects().struct
        assert _costs_of(t, "r", names_of(ing, "region", ["A", "B", "C"])) == {"A": {"B": 1}, "B": {"A": 2}, "C": {"A": 3}}
        assert _costs_of(t, "z", names_of(ing, "zone", ["Z1", "Z2"])) == {"Z1": {"Z2": 10}, "Z2": {"Z1": 10}}
        # a duplicated origin among more than 12 manual entries: Go's sort.Sort is unstable there — refused
        many = [(f"o{i:02d}", [("x", i)]) for i in range(12)] + [("o03", [("x", 99)])]
        with pytest.raises(Exception):
            ing.feed_nettopo(cr("UserDefined", [tl(REGION, many)]), "UserDefined")


def test_selector_ids_stay_lexicographic_whatever_the_feed_order(hdr):
    """FindPodOrder compares selector strings (util.go:138-153), so selector ids must follow the strings' order — also when a
    caller-seeded table lacks a selector, and when pods (which intern selectors first-seen) arrive before the AppGroup CRs"""
    with NrtIngest(["n0"]) as ing:
        ing.seed("selector", ["p2", "p1"])             # out of order, and lacks p3
        ing.feed_appgroups(json.dumps(appgroup_cr("basic", GN.APPGROUP_BASIC)).encode())
        assert [ing.name_id("selector", s) for s in ("p1", "p2", "p3")] == [0, 1, 2]
        with pytest.raises(ValueError, match="selector"):
            ing.feed_appgroups(b'{"metadata": {"name": "g"}, "spec": {"workloads": [{"dependencies": []}]}}')


def test_pods_before_appgroups_and_reingest(hdr):
    """group rows are indexed by the AppGroup name id the pods carry; a re-ingested CR replaces its row; a group known only from
    pod labels has an empty row"""
    def pod(group, sel):
        return {"metadata": {"namespace": "default", "labels": {"appgroup.diktyo.x-k8s.io": group, "appgroup.diktyo.x-k8s.io.workload": sel}},
                "spec": {"containers": [{"name": "c"}]}}
    ga = {"workloads": [{"selector": "aa", "dependencies": [("zz", 7)]}, {"selector": "zz", "dependencies": []}], "topology_order": [("aa", 2), ("zz", 1)]}
    gb = {"workloads": [{"selector": "mm", "dependencies": [("aa", 3)]}], "topology_order": [("mm", 1)]}
    with NrtIngest(["n0"]) as ing:
        ing.feed_pods(json.dumps([pod("b", "zz"), pod("a", "aa"), pod("ghost", "qq")]).encode())
        assert [ing.name_id("appgroup", g) for g in ("b", "a", "ghost")] == [0, 1, 2]     # first seen
        ing.feed_appgroups(json.dumps([appgroup_cr("a", ga), appgroup_cr("b", gb)]).encode())  # CRs in the other order
        sel = {s: ing.name_id("selector", s) for s in ("aa", "mm", "qq", "zz")}
        assert sel == {"aa": 0, "mm": 1, "qq": 2, "zz": 3}
        pods = ing.pod_objects().struct
        assert col(pods, "appgroup", 3) == [0, 1, 2] and col(pods, "selector", 3) == [sel["zz"], sel["aa"], sel["qq"]]   # renumbered
        t = ing.appgroup_objects().struct
        assert t.n_groups == 3
        # row 0 = group "b" (the id pods[0] carries), row 1 = group "a", row 2 = "ghost" without a CR
        assert col(t, "wl_ptr", 4) == [0, 1, 3, 3]
        assert col(t, "wl_selector", 3) == [sel["mm"], sel["aa"], sel["zz"]]
        assert col(t, "dep_selector", t.dep_ptr[3]) == [sel["aa"], sel["zz"]] and col(t, "dep_max_cost", t.dep_ptr[3]) == [3, 7]
        assert col(t, "topo_ptr", 4) == [0, 1, 3, 3]
        # the same CR again with a change: same row, no duplicate
        gb2 = {"workloads": [{"selector": "mm", "dependencies": [("aa", 9)]}], "topology_order": [("mm", 1)]}
        ing.feed_appgroups(json.dumps(appgroup_cr("b", gb2)).encode())
        t = ing.appgroup_objects().struct
        assert t.n_groups == 3 and col(t, "dep_max_cost", t.dep_ptr[3]) == [9, 7]
        # a later pod with a new, lexicographically earlier selector renumbers again; the tables follow
        ing.feed_pods(json.dumps(pod("a", "00")).encode())
        assert ing.name_id("selector", "00") == 0 and ing.name_id("selector", "aa") == 1
        t, pods = ing.appgroup_objects().struct, ing.pod_objects().struct
        assert col(t, "wl_selector", 3) == [2, 1, 4] and col(pods, "selector", 4) == [4, 1, 3, 0]


@pytest.mark.parametrize("case", GN.SCORE_CASES, ids=lambda c: f"L{c['line']}")
def test_decoded_crs_reproduce_the_reference_scores(hdr, oracle, case):
    """networkoverhead_test.go:572-818 end to end from wire formats: Node, Pod, AppGroup and NetworkTopology JSON -> decoder ->
    oracle NetworkOverhead.Score + NormalizeScore == the reference's expected lists (the scheduled-pods list, which no CR
    carries, comes from the Python builder)"""
    node_docs = [{"metadata": {"name": n, "labels": {REGION: r, ZONE: z}}, "status": {"allocatable": {"cpu": "8", "memory": "16Gi"}, "capacity": {"cpu": "8"}}}
                 for n, r, z in GN.NODES]
    pod = {"metadata": {"namespace": "default", "labels": {"appgroup.diktyo.x-k8s.io": "basic", "appgroup.diktyo.x-k8s.io.workload": case["selector"]}},
           "spec": {"containers": [{"name": "c"}]}}
    with NrtIngest([n for n, _, _ in GN.NODES]) as ing:
        ing.feed_appgroups(json.dumps(appgroup_cr("basic", GN.APPGROUP_BASIC)).encode())
        ing.feed_nodes(json.dumps(node_docs).encode())
        ing.feed_nettopo(json.dumps(nettopo_cr(GN.REGION_COSTS, GN.ZONE_COSTS)).encode(), "UserDefined")
        ing.feed_pods(json.dumps(pod).encode())
        # placed pods: GetScheduledList reads them from the pod lister; rebuild the group table with them through the builder
        sel = O.Interner()
        for s in ("p1", "p2", "p3"):
            sel.id(s)
        sel.freeze_sorted()
        assert all(ing.name_id("selector", s) == i for s, i in sel.ids.items())
        g = dict(GN.APPGROUP_BASIC)
        g["topology_order"] = sorted(g["topology_order"])
        g["placed"] = GN.SCORE_PLACED
        ag = O.build_appgroup_objects(hdr, sel, [g], {n: i for i, (n, _, _) in enumerate(GN.NODES)})
        # as in tests/test_oracle_golden_network.py: the reference test scores every node and normalises the full list, so the
        # plugin's own Filter is not applied (networkoverhead_test.go:790-814)
        import ctypes as C
        n = len(GN.NODES)
        sat, vio, cost = (np.zeros(n, np.int64) for _ in range(3))
        i64p = C.POINTER(C.c_int64)
        oracle.lib().orc_net_prefilter(ing.node_objects().ref(), ing.pod_objects().ref(), ag.ref(), ing.nettopo_objects().ref(), 0,
                                       sat.ctypes.data_as(i64p), vio.ctypes.data_as(i64p), cost.ctypes.data_as(i64p))
        assert cost.tolist() == case["before"]
        oracle.lib().orc_net_normalize(cost.ctypes.data_as(i64p), n)
        assert cost.tolist() == case["after"]
