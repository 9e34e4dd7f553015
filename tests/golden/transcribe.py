#!/usr/bin/env python3
"""Transcribes the reference's table-driven Go tests into JSON fixtures (tests/golden/*.json).

Runs only where /root/reference is mounted (the build container); the JSON it writes is committed and is
what the tests read — nothing under tests/ touches /root/reference at run time.  It reads composite
literals only (goparse.py); no reference code is executed (there is no Go toolchain here).

usage: python tests/golden/transcribe.py            # rewrites every fixture
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from goparse import Call, Func, Ident, line_of, parse_literal_after  # noqa: E402

REF = Path("/root/reference")

NAMES = {  # identifiers used as resource names in the test files (filter_test.go:39-48, score_test.go, least_numa_test.go)
    "v1.ResourceCPU": "cpu", "v1.ResourceMemory": "memory", "cpu": "cpu", "memory": "memory",
    "v1.ResourceEphemeralStorage": "ephemeral-storage", "v1.ResourceStorage": "storage", "v1.ResourcePods": "pods",
    "extended": "namespace/extended", "hugepages2Mi": "hugepages-2Mi", "nicResourceName": "vendor/nic1",
    "notExistingNICResourceName": "vendor/notexistingnic", "nicResourceNameNoNUMA": "vendor.com/old-nic-model",
    "gpuResource": "gpu", "gpu": "gpu",  # least_numa_test.go:32, score_test.go:42
}


def const(src: str, name: str) -> str:
    import re
    m = re.search(r"\b%s\s*=\s*\"([^\"]+)\"" % name, src)
    return m.group(1)


def rname(k):
    if isinstance(k, Ident):
        k = k.name
    return NAMES.get(k, k)


def qty(v):
    """resource.MustParse("2Gi") -> "2Gi"; *resource.NewQuantity(2, DecimalSI) -> 2; NewMilliQuantity(n) -> "<n>m"."""
    if isinstance(v, Call):
        if v.fn == "resource.MustParse":
            return v.args[0]
        if v.fn == "resource.NewQuantity":
            return v.args[0]
        if v.fn == "resource.NewMilliQuantity":
            return f"{v.args[0]}m"
    if isinstance(v, (int, str)):
        return v
    raise ValueError(f"cannot read quantity {v!r}")


def rlist(d, names):
    return {names(k): qty(v) for k, v in d.items()}


def zones_of(z_list, names):
    out = []
    for z in z_list:
        rs = [[names(c.args[0]), c.args[1], c.args[2]] for c in z.get("Resources", [])]
        costs = [[c["Name"], c["Value"]] for c in z.get("Costs", [])]
        out.append({"name": z["Name"], "type": z["Type"], "resources": rs, "costs": costs})
    return out


def nrt_of(n, names):
    pol = n.get("TopologyPolicies", [])
    pols = []
    for p in pol:
        inner = p.args[0] if isinstance(p, Call) else p
        pols.append(inner.name.split(".")[-1] if isinstance(inner, Ident) else inner)
    attrs = {a["Name"]: a["Value"] for a in n.get("Attributes", [])}
    name = n["ObjectMeta"]["Name"]
    return {"name": name, "policies": pols, "attributes": attrs, "zones": zones_of(n["Zones"], names)}


def pod_of(expr, descs, names):
    """The pod builders of objects.go:26-140 / filter_test.go:1206-1243."""
    def resolve(v):
        if isinstance(v, Call) and v.fn == "findAvailableResourceByName":  # filter_test.go:1197-1204
            sel = v.args[0]  # nodeTopologyDescs[i].nrt.Zones[j].Resources
            zi = sel.args[0].args[1]
            di = sel.args[0].args[0].args[0].args[0].args[1]
            want = names(v.args[1])
            for n, _cap, av in descs[di]["zones"][zi]["resources"]:
                if n == want:
                    return av
            raise KeyError(want)
        return qty(v)

    def rl(d):
        return {names(k): resolve(v) for k, v in d.items()}

    if isinstance(expr, dict) or (isinstance(expr, list) and not expr):  # &v1.Pod{}
        return {"containers": []}
    if expr.fn == "makePodByResourceList":
        r = rl(expr.args[0])
        return {"containers": [{"requests": r, "limits": r}]}
    if expr.fn == "makePodByResourceListWithManyContainers":  # objects.go:104-120
        r = rl(expr.args[0])
        return {"containers": [{"requests": r, "limits": r} for _ in range(expr.args[1])]}
    if expr.fn == "makePodWithReqByResourceList":
        return {"containers": [{"requests": rl(expr.args[0])}]}
    if expr.fn == "makePodWithReqAndLimitByResourceList":
        return {"containers": [{"requests": rl(expr.args[0]), "limits": rl(expr.args[1])}]}
    if expr.fn == "makePod":
        pod = {"containers": [], "init_containers": []}
        for opt in expr.args[1:]:
            lists = [rl(x) for x in opt.args[0]]
            if opt.fn == "withMultiContainers":  # requests == limits (filter_test.go:1218-1234)
                pod["containers"] += [{"requests": r, "limits": r} for r in lists]
            elif opt.fn == "withMultiInitContainers":
                pod["init_containers"] += [{"requests": r, "limits": r} for r in lists]
        return pod
    raise ValueError(expr)


def status_of(v):
    if isinstance(v, Ident) and v.name == "nil":
        return None
    if isinstance(v, Call) and v.fn == "fwk.NewStatus":
        return {"code": v.args[0].name.split(".")[-1], "message": v.args[1]}
    raise ValueError(v)


def nrt_filter():
    path = "pkg/noderesourcetopology/filter_test.go"
    src = (REF / path).read_text()
    names = rname
    out = {"source": path, "note": "TestNodeResourceTopology / ...MultiContainerPodScope / ...MultiContainerContainerScope"}
    descs_raw = parse_literal_after(src, "nodeTopologyDescs := ")
    descs = []
    for d in descs_raw:
        n = nrt_of(d["nrt"], names)
        n["node_extra"] = rlist(d.get("node", {}), names)
        descs.append(n)
    out["nodes"] = descs
    cases = []
    pos = src.index("tests := ")
    for t in parse_literal_after(src, "tests := "):
        cases.append({"name": t["name"], "line": line_of(src, '"' + t["name"] + '"', pos), "pod": pod_of(t["pod"], descs, names),
                      "node": t["node"].args[1], "want": status_of(t["wantStatus"])})
    out["cases"] = cases
    # pod scope, multi container (filter_test.go:715-941)
    p2 = src.index("func TestNodeResourceTopologyMultiContainerPodScope")
    nts = parse_literal_after(src[p2:], "nodeTopologies := ")
    out["pod_scope_nodes"] = [nrt_of(n, names) for n in nts]
    ps = []
    for t in parse_literal_after(src[p2:], "tests := "):
        ps.append({"name": t["name"], "line": line_of(src, '"' + t["name"] + '"', p2), "pod": pod_of(t["pod"], descs, names),
                   "node": t["node"].args[1], "want": status_of(t["wantStatus"])})
    out["pod_scope_cases"] = ps
    # container scope, tiered cases (filter_test.go:943-1182)
    p3 = src.index("func TestNodeResourceTopologyMultiContainerContainerScope")
    nts = parse_literal_after(src[p3:], "nodeTopologies := ")
    out["container_scope_nodes"] = [nrt_of(n, names) for n in nts]
    cs = []
    for t in parse_literal_after(src[p3:], "tue := "):
        def rls(lst):
            return [{"requests": {names(k): v for k, v in m.items()}, "limits": {names(k): v for k, v in m.items()}} for m in lst]
        cs.append({"name": t["description"], "line": line_of(src, '"' + t["description"] + '"', p3),
                   "pod": {"init_containers": rls(t.get("initCntReq", [])), "containers": rls(t.get("cntReq", []))},
                   "node": 0, "want": ({"code": "Unschedulable", "message": t["statusErr"]} if t.get("statusErr") else None)})
    out["container_scope_cases"] = cs
    return out


def nrt_score():
    path = "pkg/noderesourcetopology/score_test.go"
    src = (REF / path).read_text()
    names = rname
    out = {"source": path}
    pd = src.index("func defaultNUMANodes")
    out["default_numa_nodes"] = [nrt_of(n, names) for n in parse_literal_after(src[pd:], "nrts := ")]
    pf = src.index("func fourNUMANodes")
    out["four_numa_nodes"] = [nrt_of(n, names) for n in parse_literal_after(src[pf:], "return ")]
    # TestNodeResourceScorePlugin (:88-195) and TestNodeResourcePartialDataScorePlugin (:484-621)
    def scen(fn_name):
        p = src.index("func " + fn_name)
        reqs = parse_literal_after(src[p:], "pRequests := ")
        pod = pod_of(reqs[0]["pod"], [], names)
        res = []
        for t in parse_literal_after(src[p:], "tests := "):
            f = t.get("nrtFilter")
            keep = None
            if isinstance(f, Func):
                keep = [] if 'nrt.Name != "Node1"' not in f.src else ["Node1"]
            res.append({"name": t["name"], "line": line_of(src, '"' + t["name"] + '"', p), "pod": pod,
                        "strategy": {"mostAllocatedScoreStrategy": "MostAllocated", "leastAllocatedScoreStrategy": "LeastAllocated",
                                     "balancedAllocationScoreStrategy": "BalancedAllocation"}[t["strategy"].name],
                        "wanted": t["wantedRes"], "nodes_with_nrt": keep})
        return res
    out["strategy_cases"] = scen("TestNodeResourceScorePlugin")
    out["partial_data_cases"] = scen("TestNodeResourcePartialDataScorePlugin")
    p = src.index("func TestNodeResourceScorePluginLeastNUMA")
    ln = []
    for t in parse_literal_after(src[p:], "testCases := "):
        nodes = t["nodes"]
        fixture = {"fn": nodes.fn, "policy": None}
        if nodes.fn == "defaultNUMANodes":
            fixture["policy"] = nodes.args[0].args[0].name.split(".")[-1]
        ln.append({"name": t["name"], "line": line_of(src, '"' + t["name"] + '"', p),
                   "containers": [{names(k): qty(v) for k, v in rl.items()} for rl in t["podRequests"]],
                   "wanted": t["wantedRes"], "nodes": fixture})
    out["least_numa_cases"] = ln
    return out


def nrt_least_numa():
    path = "pkg/noderesourcetopology/least_numa_test.go"
    src = (REF / path).read_text()
    names = rname
    out = {"source": path}
    cases = []
    for t in parse_literal_after(src, "testCases := "):
        bm = t["expectedBitmask"]
        cases.append({
            "name": t["description"], "line": line_of(src, '"' + t["description"] + '"'),
            "numa_nodes": [{"id": n["NUMAID"], "resources": {names(k): qty(v) for k, v in n.get("Resources", {}).items()},
                            "costs": {str(k): v for k, v in n.get("Costs", {}).items()}} for n in t["numaNodes"]],
            "pod_resources": {names(k): qty(v) for k, v in t["podResources"].items()},
            "bitmask": None if isinstance(bm, Ident) else list(bm.args),
            "min_distance": t["expectedMinDistance"].name == "true",
        })
    out["numa_nodes_required"] = cases
    p = src.index("func TestNormalizeScore")
    out["normalize_score"] = [{"name": t["description"], "count": t["score"], "expected": t["expectedScore"],
                               "optimal": isinstance(t.get("optimalDistance"), Ident) and t["optimalDistance"].name == "true"}
                              for t in parse_literal_after(src[p:], "tcases := ")]
    return out


def capacity():
    """pkg/capacityscheduling/elasticquota_test.go: TestUsedOverMinWith (:133), TestUsedOverMaxWith (:265), TestUsedOverMin (:397)"""
    path = "pkg/capacityscheduling/elasticquota_test.go"
    src = (REF / path).read_text()

    def fres(d):
        if d is None or isinstance(d, Ident):
            return None
        out = {k: v for k, v in d.items() if k != "ScalarResources"}
        if "ScalarResources" in d:
            out["ScalarResources"] = {("nvidia.com/gpu" if k == "ResourceGPU" else k): v for k, v in d["ScalarResources"].items()}
        return out

    out = {"source": path}
    for fn, key, bound in (("TestUsedOverMinWith", "used_over_min_with", "Min"), ("TestUsedOverMaxWith", "used_over_max_with", "Max"),
                           ("TestUsedOverMin", "used_over_min", "Min")):
        p = src.index("func " + fn + "(")
        cases = []
        for t in parse_literal_after(src[p:], "tests := "):
            b = t["before"]
            cases.append({"name": t["name"], "line": line_of(src, '"' + t["name"] + '"', p), "used": fres(b.get("Used")),
                          "bound": fres(b.get(bound)), "pod_request": fres(t.get("podRequest")),
                          "expected": t["expected"].name == "true"})
        out[key] = cases
    return out


_NRT_CONSTS = {"cpu": "cpu", "memory": "memory", "gpuResourceName": "vendor/gpu", "hugepages2Mi": "hugepages-2Mi",
               "nicResourceName": "vendor/nic1", "v1.ResourceCPU": "cpu", "v1.ResourceMemory": "memory",
               "corev1.ResourceCPU": "cpu", "corev1.ResourceMemory": "memory"}
_NRT_ATTRS = {"nodeconfig.AttributePolicy": "topologyManagerPolicy", "nodeconfig.AttributeScope": "topologyManagerScope"}


def _make_nrt(expr):
    """a MakeNRT() builder chain (test/integration/nrtutils.go:159-207) as data"""
    def rn(k):
        return _NRT_CONSTS[k.name] if isinstance(k, Ident) else _NRT_CONSTS.get(k, k)
    out = {"name": None, "policies": [], "attributes": {}, "zones": []}

    def walk(e):
        if e.fn == "MakeNRT":
            return
        walk(e.args[0])
        if e.fn == ".Name":
            out["name"] = e.args[1]
        elif e.fn == ".Policy":
            out["policies"].append(e.args[1].name.split(".")[-1])
        elif e.fn == ".Attributes":
            for a in e.args[1]:
                out["attributes"][_NRT_ATTRS[a["Name"].name]] = a["Value"]
        elif e.fn == ".Zone":
            out["zones"].append({"name": f"node-{len(out['zones'])}", "type": "Node",
                                 "resources": [[rn(r.args[0]), r.args[1], r.args[2]] for r in e.args[1]]})
        elif e.fn != ".Obj":
            raise ValueError(f"unknown NRT builder {e.fn}")
    walk(expr)
    return out


def nrt_cache_integration():
    """test/integration/noderesourcetopology_cache_test.go:111-644, TestTopologyCachePluginWithoutUpdates (6 cases): pods created one
    after the other against two nodes whose NRT objects never change; what the second pod sees depends on the cache —
    OverReserve (default profile; overreserve.go:170-203: a bound pod's request is charged to EVERY zone of its node, deletes are
    ignored until a resync) or DiscardReserved (profile "discardReserved": nothing is charged once PostBind ran).  Steps are
    recorded in order: a pod (one container per resources map; util.WithLimits = limits, which the API server copies into
    requests; util.WithRequests = requests only) with the node it must land on ("" = stays pending, "*" = any), or a delete.
    Nodes are created from the NRTs: capacity = the sum of the zones' capacities per resource, pods "128" (nrtutils.go:73-90,
    :240-253).  makeTestFullyAvailableNRTs() (nrtutils.go:278-311) is read from its own literal.  The two other tests of the file
    (resync after pod-fingerprint / attribute updates) are not transcribed."""
    path = "test/integration/noderesourcetopology_cache_test.go"
    src = (REF / path).read_text()
    utils = (REF / "test/integration/nrtutils.go").read_text()
    single = parse_literal_after(utils[utils.index("func makeTestFullyAvailableNRTSingle"):], "return ")
    second = parse_literal_after(utils[utils.index("func makeTestFullyAvailableNRTs"):], "return ")
    assert isinstance(second, Call) and second.fn == "append" and isinstance(second.args[0], Ident) and second.args[0].name == "nrts"
    fully_available = [_make_nrt(n) for n in single] + [_make_nrt(n) for n in second.args[1:]]
    p = src.index("func TestTopologyCachePluginWithoutUpdates")
    end = src.index("func TestTopologyCachePluginWithPodFingerprintUpdates")
    cases = []
    cursor = p
    for t in parse_literal_after(src[p:end], "range []testCase"):
        cursor = src.index('"' + t["name"] + '"', cursor) + 1
        nrts = t["nodeResourceTopologies"]
        if isinstance(nrts, Call):
            assert nrts.fn == "makeTestFullyAvailableNRTs", nrts.fn
            nrts = fully_available
        else:
            nrts = [_make_nrt(n) for n in nrts]
        steps, cache = [], "OverReserve"
        for d in t["podDescs"]:
            sched = d.get("schedulerName")
            sched = {"discardReservedSchedulerName": "discardReserved"}.get(sched.name, sched.name) if isinstance(sched, Ident) else sched
            if sched == "discardReserved":
                cache = "DiscardReserved"
            if isinstance(d.get("isDelete"), Ident) and d["isDelete"].name == "true":
                steps.append({"delete": d["podName"]})
                continue
            guaranteed = d["isGuaranteed"].name == "true"
            maps = ([d["resourcesMap"]] if d.get("resourcesMap") else []) + list(d.get("multiResourcesMap") or [])
            ctrs = []
            for m in maps:
                rl = {_NRT_CONSTS.get(k, k): v for k, v in m.items()}
                ctrs.append({"requests": dict(rl), "limits": dict(rl)} if guaranteed else {"requests": dict(rl)})
            exp = d["expectedNode"]
            exp = {"anyNode": "*"}[exp.name] if isinstance(exp, Ident) else exp
            steps.append({"pod": d["podName"], "containers": ctrs, "expected_node": exp})
        cases.append({"name": t["name"], "line": src.count("\n", 0, cursor) + 1, "cache": cache, "strategy": "LeastAllocated", "nrts": nrts, "steps": steps})
    return {"source": path, "node_extra_capacity": {"pods": "128"}, "cases": cases}


def nrt_integration():
    """test/integration/noderesourcetopology_test.go: the table of TestTopologyMatchPlugin (29 cases): one pod, two nodes
    with NRTs built by the MakeNRT() wrapper (test/integration/nrtutils.go:159-207), the scheduler profile (= scoring
    strategy) picked by SchedulerName, and the nodes the pod may land on.  Containers built with util.WithLimits carry
    limits only; the API server the integration test talks to defaults requests to limits, so both are recorded."""
    path = "test/integration/noderesourcetopology_test.go"
    src = (REF / path).read_text()
    consts = {"cpu": "cpu", "memory": "memory", "gpuResourceName": "vendor/gpu", "hugepages2Mi": "hugepages-2Mi",
              "nicResourceName": "vendor/nic1", "v1.ResourceCPU": "cpu", "v1.ResourceMemory": "memory"}
    strategy_of = {None: "MostAllocated", "mostAllocatedScheduler": "MostAllocated", "balancedAllocationScheduler": "BalancedAllocation",
                   "leastAllocatedScheduler": "LeastAllocated", "leastNUMAScheduler": "LeastNUMANodes"}
    attr_names = {"nodeconfig.AttributePolicy": "topologyManagerPolicy", "nodeconfig.AttributeScope": "topologyManagerScope"}

    def rn(k):
        return consts[k.name] if isinstance(k, Ident) else consts.get(k, k)

    def rl(d):
        return {rn(k): v for k, v in d.items()}

    def pod_of(expr):
        pod = {"containers": [], "init_containers": [], "scheduler": None}

        def walk(e):
            if isinstance(e, Call):
                if e.fn == "st.MakePod":
                    return
                walk(e.args[0])
                if e.fn == "util.WithLimits":
                    init = isinstance(e.args[2], Ident) and e.args[2].name == "true"
                    lim = rl(e.args[1])
                    if lim:
                        (pod["init_containers"] if init else pod["containers"]).append({"requests": dict(lim), "limits": dict(lim)})
                elif e.fn == ".Req":
                    pod["containers"].append({"requests": rl(e.args[1])})
                elif e.fn == ".Container":
                    pod["containers"].append({})
                elif e.fn == ".SchedulerName":
                    pod["scheduler"] = e.args[1].name
                elif e.fn in (".Namespace", ".Name", ".Obj"):
                    pass
                else:
                    raise ValueError(f"unknown pod builder {e.fn}")
        walk(expr)
        return pod

    def nrt_of(expr):
        out = {"name": None, "policies": [], "attributes": {}, "zones": []}

        def walk(e):
            if e.fn == "MakeNRT":
                return
            walk(e.args[0])
            if e.fn == ".Name":
                out["name"] = e.args[1]
            elif e.fn == ".Policy":
                out["policies"].append(e.args[1].name.split(".")[-1])
            elif e.fn == ".Attributes":
                for a in e.args[1]:
                    out["attributes"][attr_names[a["Name"].name]] = a["Value"]
            elif e.fn in (".Zone", ".ZoneWithCosts"):
                z = {"name": f"node-{len(out['zones'])}", "type": "Node",
                     "resources": [[rn(r.args[0]), r.args[1], r.args[2]] for r in e.args[1]]}
                if e.fn == ".ZoneWithCosts":
                    z["costs"] = {c["Name"]: c["Value"] for c in e.args[2]}
                out["zones"].append(z)
            elif e.fn != ".Obj":
                raise ValueError(f"unknown NRT builder {e.fn}")
        walk(expr)
        return out

    cases = []
    cursor = 0
    for t in parse_literal_after(src, "tests := "):
        assert len(t["pods"]) == 1
        pod = pod_of(t["pods"][0])
        cursor = src.index('"' + t["name"] + '"', cursor) + 1  # names repeat: search onwards from the previous case
        cases.append({"name": t["name"], "line": src.count("\n", 0, cursor) + 1, "strategy": strategy_of[pod.pop("scheduler")], "pod": pod,
                      "nrts": [nrt_of(n) for n in t.get("nodeResourceTopologies", [])], "expected_nodes": list(t["expectedNodes"])})
    return {"source": path, "node_names": ["fake-node-1", "fake-node-2"],
            "node_capacity": {"cpu": "64", "memory": "128Gi", "pods": "32", "hugepages-2Mi": "896Mi", "vendor/nic1": "48", "ephemeral-storage": "32Gi"},
            "cases": cases}


def trimaran_handler():
    """pkg/trimaran/handler_test.go:12-77 TestHandlerCacheCleanup: the PodAssignEventHandler cache after OnUpdate + cleanupCache.
    Timestamps become offsets in seconds from time.Now() (null: the zero time.Time of an entry built without one)."""
    import re
    path = "pkg/trimaran/handler_test.go"
    src = (REF / path).read_text()
    pod_names = dict(re.findall(r"(pod\d) := st\.MakePod\(\)\.Name\(\"([^\"]+)\"\)", src))
    unit = {"time.Minute": 60, "time.Second": 1}

    def offset(ts):
        if ts is None:
            return None
        if isinstance(ts, Call) and ts.fn == "time.Now":
            return 0
        assert isinstance(ts, Call) and ts.fn == ".Add" and ts.args[0].fn == "time.Now" and ts.args[1].fn == "op*", ts
        n, u = ts.args[1].args
        return n * unit[u.name]

    def name_of(v):
        return pod_names[v.name.split(".")[0]] if isinstance(v, Ident) else v

    cases = []
    for t in parse_literal_after(src, "tests := "):
        cases.append({"name": t["name"], "line": line_of(src, '"' + t["name"] + '"', 0),
                      "cache": [{"pod": name_of(e["Pod"]), "age_offset_s": offset(e.get("Timestamp"))} for e in t["podInfoList"]],
                      "pod_to_update": t.get("podToUpdate", ""), "expected_pods": [name_of(v) for v in t["expectedCachePods"]],
                      "expected_size": t["expectedCacheSize"]})
    return {"source": path, "node": re.search(r'testNode := "([^"]+)"', src).group(1), "reporting_interval_s": 60, "cases": cases}


def nrt_discard_reserved():
    """pkg/noderesourcetopology/cache/discardreserved_test.go:34-140: the four tests of the DiscardReserved cache as operation lists.
    The tests are straight-line code, not tables: the operations and assertions below are read off the statements (cited per step)."""
    import re
    path = "pkg/noderesourcetopology/cache/discardreserved_test.go"
    src = (REF / path).read_text()

    def ln(needle, start=0):
        return line_of(src, needle, start)

    out = {"source": path, "tests": []}
    # TestDiscardReservedNodesGetCachedNRTCopy (:34-58): one table case through checkGetCachedNRTCopy
    p = src.index("func TestDiscardReservedNodesGetCachedNRTCopy")
    node = re.search(r'testNodeName := "([^"]+)"', src[p:]).group(1)
    (case,) = parse_literal_after(src[p:], "testCases := ")
    out["tests"].append({"name": "TestDiscardReservedNodesGetCachedNRTCopy", "line": ln("func TestDiscardReservedNodesGetCachedNRTCopy"),
                         "steps": [{"op": "store_nrt", "node": node},
                                   {"op": "get", "node": node, "has_foreign_pods": case["hasForeignPods"].name == "true",
                                    "expect_ok": case["expectedOK"].name == "true", "expect_nrt": True, "case": case["name"]}]})
    # TestDiscardReservedNodesGetNRTCopyFails (:60-77): a reservation on node1 -> (nil, Fresh false)
    p = src.index("func TestDiscardReservedNodesGetNRTCopyFails")
    m = re.search(r'"(node\d)": \{\s*types\.UID\("([^"]+)"\): true', src[p:])
    out["tests"].append({"name": "TestDiscardReservedNodesGetNRTCopyFails", "line": ln("func TestDiscardReservedNodesGetNRTCopyFails"),
                         "steps": [{"op": "preset", "node": m.group(1), "uid": m.group(2)},
                                   {"op": "get", "node": re.search(r'GetCachedNRTCopy\(context\.Background\(\), "([^"]+)"', src[p:]).group(1),
                                    "expect_ok": False, "expect_nrt": False}]})
    # TestDiscardReservedNodesReserveNodeResources (:79-104)
    p = src.index("func TestDiscardReservedNodesReserveNodeResources")
    m = re.search(r'ReserveNodeResources\("([^"]+)", &corev1\.Pod\{\s*ObjectMeta: metav1\.ObjectMeta\{\s*Name:\s*"([^"]+)",\s*Namespace:\s*"([^"]+)",\s*UID:\s*"([^"]+)"', src[p:])
    out["tests"].append({"name": "TestDiscardReservedNodesReserveNodeResources", "line": ln("func TestDiscardReservedNodesReserveNodeResources"),
                         "steps": [{"op": "reserve", "node": m.group(1), "uid": m.group(4)},
                                   {"op": "expect_map", "node": m.group(1), "uids": {m.group(4): True}}]})
    # TestDiscardReservedNodesRemoveReservationForNode (:106-150)
    p = src.index("func TestDiscardReservedNodesRemoveReservationForNode")
    uid = re.search(r'UID:\s*"([^"]+)"', src[p:]).group(1)
    n1 = re.search(r'ReserveNodeResources\("([^"]+)", pod\)', src[p:]).group(1)
    n2 = re.search(r'removeReservationForNode\("([^"]+)", pod\)', src[p:]).group(1)
    out["tests"].append({"name": "TestDiscardReservedNodesRemoveReservationForNode", "line": ln("func TestDiscardReservedNodesRemoveReservationForNode"),
                         "steps": [{"op": "reserve", "node": n1, "uid": uid}, {"op": "expect_map", "node": n1, "uids": {uid: True}},
                                   {"op": "remove", "node": n2, "uid": uid}, {"op": "expect_map", "node": n2, "uids": {}}]})
    return out


FIXTURES = {"capacity.json": capacity, "nrt_filter.json": nrt_filter, "nrt_score.json": nrt_score, "nrt_least_numa.json": nrt_least_numa,
            "nrt_integration.json": nrt_integration, "nrt_cache_integration.json": nrt_cache_integration, "trimaran_handler.json": trimaran_handler, "nrt_discard_reserved.json": nrt_discard_reserved}

if __name__ == "__main__":
    for fname, fn in FIXTURES.items():
        data = fn()
        (HERE / fname).write_text(json.dumps(data, indent=1, sort_keys=False) + "\n")
        print(fname, {k: (len(v) if isinstance(v, list) else "…") for k, v in data.items()})
