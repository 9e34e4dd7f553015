#!/usr/bin/env python3
"""Checks hand-typed golden tables against the reference's Go test tables, read with goparse (no Go toolchain here, nothing is
executed).  Runs only where /root/reference is mounted; tests/test_golden_hand_typed.py calls it and is skipped elsewhere.
Covered: allocatable.py (pkg/noderesources/allocatable_test.go:114-238) — every case's pod, node sizes, resource weights, mode,
expected list and line; trimaran.py's COMPUTE_SCORE and MU_SIGMA; lroc.py's three Beta-distribution tables and the tolerance;
peaks.py's power model and NormalizeScore cases.  README.md lists what is still hand-typed only."""
from __future__ import annotations

import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from goparse import Call, Ident, line_of, parse_literal_after  # noqa: E402

REF = Path("/root/reference")


def check_allocatable() -> int:
    import allocatable as A
    src = (REF / "pkg/noderesources/allocatable_test.go").read_text()
    p = src.index("func TestNodeResourcesAllocatable")
    score = {"fwk.MinNodeScore": 0, "fwk.MaxNodeScore": 100}
    sets = {"defaultResourceAllocatableSet": A.DEFAULT, "cpuResourceAllocatableSet": A.CPU_HEAVY}
    pods = {"cpuAndMemory": A.CPU_AND_MEMORY, "bigCpu": A.BIG_CPU}
    checked = 0
    hand = {c["line"]: c for c in A.CASES}
    bad = {c["line"]: c for c in A.INVALID}
    for t in parse_literal_after(src[p:], "tests := "):
        name = t["name"]
        # the hand-typed tables cite the line of a case's first field: the one after its opening brace
        at = src.index('"' + name + '"', p)
        start = src.rfind("\n\t\t{\n", 0, at) + 3
        line = src.count("\n", 0, start) + 2
        args = t["args"]
        res = args["Resources"]
        if isinstance(res, Ident):
            weights = sets[res.name]
        else:
            weights = {e["Name"]: e["Weight"] for e in res}  # (literal lists name the resources with plain strings)
        if "wantErr" in t:
            c = bad[line + 1]  # (these two open with a comment line)
            assert c["resources"] == weights and c["name"] == name, (line, weights, c)
            checked += 1
            continue
        c = hand[line]
        # (names are abbreviated in the hand-typed table — and two of the reference's own names contradict the case's Mode, :183-196 — the line ties them)
        pod = t["pod"]
        want_pod = pods[pod.name] if isinstance(pod, Ident) else A.NO_RESOURCES
        assert c["pod"] == want_pod, (line, pod)
        nodes = [(n.args[1], n.args[2]) for n in t["nodeInfos"]]
        nodes = [tuple(eval_const(v) for v in n) for n in nodes]
        assert c["nodes"] == nodes, (line, nodes, c["nodes"])
        assert c["resources"] == weights, (line, weights)
        assert c["mode"] == {"modeLeast": "Least", "modeMost": "Most"}[args["Mode"].name], line
        exp = [eval_const(e["Score"], score) for e in t["expectedList"]]
        assert c["expected"] == exp, (line, exp, c["expected"])
        checked += 1
    assert checked == len(A.CASES) + len(A.INVALID), (checked, len(A.CASES), len(A.INVALID))
    return checked


def eval_const(v, names=None):
    """integer expressions as the test writes them: 1000 * 1024 * 1024, (fwk.MinNodeScore + fwk.MaxNodeScore) / 2"""
    if isinstance(v, int):
        return v
    if isinstance(v, Ident):
        return names[v.name]
    if isinstance(v, Call) and v.fn in ("op*", "op+", "op/"):
        a = [eval_const(x, names) for x in v.args]
        out = a[0]
        for x in a[1:]:
            out = out * x if v.fn == "op*" else (out + x if v.fn == "op+" else out // x)
        return out
    raise ValueError(repr(v))


def _num(v):
    return float(v) if isinstance(v, (int, float)) else {"true": True, "false": False}[v.name]


def _tests(path: str, func: str, var: str = "tests := "):
    src = (REF / path).read_text()
    return parse_literal_after(src[src.index("func " + func):], var)


def check_trimaran() -> int:
    """trimaran.py: COMPUTE_SCORE (analysis_test.go:38-157, TestComputeScore) and MU_SIGMA (resourcestats_test.go:259-372,
    TestGetMuSigma) — every number of every case, in order (names: the hand-typed table tags the reference's repeated
    "large usedAvg" with "(dup)")"""
    import trimaran as T
    checked = 0
    go = _tests("pkg/trimaran/loadvariationriskbalancing/analysis_test.go", "TestComputeScore")
    assert len(go) == len(T.COMPUTE_SCORE)
    for t, c in zip(go, T.COMPUTE_SCORE):
        rs = t["rs"]
        want = (t["margin"], t["sensitivity"], rs["Capacity"], rs["Req"], rs["UsedAvg"], rs["UsedStdev"], t["expected"])
        assert c[0].replace(" (dup)", "") == t["name"] and tuple(c[1:]) == want, (t["name"], c, want)
        checked += 1
    go = _tests("pkg/trimaran/resourcestats_test.go", "TestGetMuSigma")
    assert len(go) == len(T.MU_SIGMA)
    for t, c in zip(go, T.MU_SIGMA):
        rs = t["args"]["rs"]
        want = (rs["Capacity"], rs["Req"], rs["UsedAvg"], rs["UsedStdev"], t["wantMu"], t["wantSigma"])
        assert c[0] == t["name"] and tuple(float(x) for x in c[1:]) == tuple(float(x) for x in want), (t["name"], c, want)
        checked += 1
    return checked


def check_trimaran_score_cases() -> int:
    """trimaran.py: TLP_CASES (targetloadpacking_test.go:148-238, TestTargetLoadPackingScoring) and LVRB_CASES
    (loadvariationriskbalancing_test.go:152-328, TestScore): per case its name and line, the pod (the empty pod or
    getPodWithContainersAndOverhead's arguments), the watcher response's metrics (nil response = the 404 case) and the expected
    score; the node's sizes; DefaultTargetUtilizationPercent from apis/config/v1/defaults.go"""
    import re
    import trimaran as T
    target = int(re.search(r"DefaultTargetUtilizationPercent int64 = (\d+)", (REF / "apis/config/v1/defaults.go").read_text()).group(1))
    assert target == T.TLP_PARAMS["target_utilization"]
    consts = {"cfgv1.DefaultTargetUtilizationPercent": target, "fwk.MinNodeScore": 0, "fwk.MaxNodeScore": 100, "mega": 1024 * 1024}
    typ = {"watcher.CPU": "CPU", "watcher.Memory": "Memory"}
    op = {"watcher.Latest": "Latest", "watcher.Average": "AVG", "watcher.Std": "STD"}

    def val(v):
        if isinstance(v, Call) and v.fn == "float64":
            return val(v.args[0])
        return eval_const(v, consts)

    def metrics_of(resp):
        if not resp:  # a zero-value WatcherMetrics: the fake watcher answers 404
            return None
        return {0: [(typ[m["Type"].name], op[m["Operator"].name], val(m["Value"])) for m in resp["Data"]["NodeMetricsMap"]["node-1"]["Metrics"]]}

    checked = 0
    src = (REF / "pkg/trimaran/targetloadpacking/targetloadpacking_test.go").read_text()
    p = src.index("func TestTargetLoadPackingScoring")
    sizes = parse_literal_after(src[p:], "nodeResources := ")
    assert {"cpu": sizes["v1.ResourceCPU"], "memory": sizes["v1.ResourceMemory"]} == T.NODE, sizes
    go = parse_literal_after(src[p:], "tests := ")
    assert len(go) == len(T.TLP_CASES)
    for t, c in zip(go, T.TLP_CASES):
        pod = t["pod"]
        want_pod = T._pod_overhead(*[eval_const(a, consts) for a in pod.args]) if pod.fn == "getPodWithContainersAndOverhead" else {"containers": []}
        assert (t["test"], want_pod, metrics_of(t["watcherResponse"]), [val(e["Score"]) for e in t["expected"]]) == (c["name"], c["pod"], c["metrics"], c["expected"]), t["test"]
        assert -2 <= c["line"] - line_of(src, '"' + t["test"] + '"', p) <= 12, (t["test"], c["line"])  # (the cited line lies inside the case)
        checked += 1
    src = (REF / "pkg/trimaran/loadvariationriskbalancing/loadvariationriskbalancing_test.go").read_text()
    p = src.index("func TestScore")
    go = parse_literal_after(src[p:], "tests := ")
    assert len(go) == len(T.LVRB_CASES)
    for t, c in zip(go, T.LVRB_CASES):
        pod = t["pod"]
        if pod.fn == "getPodWithContainersAndOverhead":  # (overhead, initCpu, initMem, []cpu, []mem)
            a = pod.args
            assert [eval_const(x, consts) for x in a[:3]] == [0, 0, 0]
            want_pod = T._lv_pod([eval_const(x, consts) for x in a[3]], [eval_const(x, consts) for x in a[4]])
        else:
            want_pod = {"containers": []}
        assert (t["test"], want_pod, metrics_of(t["watcherResponse"]), [val(e["Score"]) for e in t["expected"]]) == (c["name"], c["pod"], c["metrics"], c["expected"]), t["test"]
        assert -2 <= c["line"] - line_of(src, '"' + t["test"] + '"', p) <= 12, (t["test"], c["line"])
        checked += 1
    return checked


def check_lroc() -> int:
    """lroc.py: MATCH_MOMENTS, DISTRIBUTION_FUNCTION, MAX_VARIANCE against beta_test.go:111-171, :236-327, :329-374"""
    import lroc as L
    checked = 0
    go = _tests("pkg/trimaran/lowriskovercommitment/beta_test.go", "TestBetaDistribution_MatchMoments")
    assert len(go) == len(L.MATCH_MOMENTS)
    for t, c in zip(go, L.MATCH_MOMENTS):
        f = t["fields"] or {}
        want = (t["name"], float(t["args"]["m1"]), float(t["args"]["m2"]), _num(t["want"]),
                float(f["alpha"]) if f else None, float(f["beta"]) if f else None)
        got = (c[0], float(c[1]), float(c[2]), c[3], None if c[4] is None else float(c[4]), None if c[5] is None else float(c[5]))
        assert got == want, (got, want)
        checked += 1
    go = _tests("pkg/trimaran/lowriskovercommitment/beta_test.go", "TestBetaDistribution_DistributionFunction")
    assert len(go) == len(L.DISTRIBUTION_FUNCTION)
    for t, c in zip(go, L.DISTRIBUTION_FUNCTION):
        want = (t["name"], float(t["fields"]["alpha"]), float(t["fields"]["beta"]), float(t["args"]["x"]), float(t["want"]))
        assert (c[0],) + tuple(float(x) for x in c[1:]) == want, (c, want)
        checked += 1
    go = _tests("pkg/trimaran/lowriskovercommitment/beta_test.go", "TestGetMaxVariance")
    assert [(float(t["args"]["m1"]), float(t["want"])) for t in go] == [(float(a), float(b)) for a, b in L.MAX_VARIANCE]
    checked += len(go)
    src = (REF / "pkg/trimaran/lowriskovercommitment/beta_test.go").read_text()
    tol = parse_literal_after(src, "tolerance = ") if "tolerance = " in src else None
    assert tol is None or float(tol) == L.TOLERANCE, tol
    return checked


def check_lroc_compute_risk() -> int:
    """lroc.py: the computeRisk fixtures (lowriskovercommitment_test.go:261-338: node_A's sizes, watcherData_A's metrics, nrla_A1 / nrla_A2)
    and cases (:341-395)"""
    import lroc as L
    src = (REF / "pkg/trimaran/lowriskovercommitment/lowriskovercommitment_test.go").read_text()
    res = parse_literal_after(src, "nodeResources_A map[v1.ResourceName]string = ")
    assert {"cpu": res["v1.ResourceCPU"], "memory": res["v1.ResourceMemory"]} == L.NODE_A, res
    wd = parse_literal_after(src, "watcherData_A watcher.Data = ")
    metrics = next(iter(wd["NodeMetricsMap"].values()))["Metrics"]
    typ = {"watcher.CPU": "CPU", "watcher.Memory": "Memory"}
    op = {"watcher.Average": "AVG", "watcher.Std": "STD"}
    assert [(typ[m["Type"].name], op[m["Operator"].name], m["Value"]) for m in metrics] == L.METRICS_A, metrics
    key = {"NodeRequest": "req", "NodeLimit": "lim", "NodeRequestMinusPod": "req_minus_pod", "NodeLimitMinusPod": "lim_minus_pod", "Nodecapacity": "cap"}
    fixtures = {}
    for name, hand in (("nrla_A1", L.NRLA_A1), ("nrla_A2", L.NRLA_A2)):
        lit = parse_literal_after(src, name + " *trimaran.NodeRequestsAndLimits = ")
        flat = {}
        for k, v in lit.items():
            flat[key[k] + "_cpu"], flat[key[k] + "_mem"] = v["MilliCPU"], v["Memory"]
        assert flat == hand, (name, flat, hand)
        fixtures[name] = hand
    go = parse_literal_after(src[src.index("func TestLowRiskOverCommitment_computeRisk"):], "tests := ")
    assert len(go) == len(L.COMPUTE_RISK)
    for t, c in zip(go, L.COMPUTE_RISK):
        assert (t["name"], typ[t["resourceType"].name], float(t["want"])) == (c[0], c[1], float(c[3])) and fixtures[t["nodeRequestsAndLimits"].name] is c[2], (t, c)
    return 3 + len(go)


def check_network() -> int:
    """network.py: TestNetworkOverheadScore's pod, raw scores and normalised scores per node (networkoverhead_test.go:572-818, 3 cases) and
    TestNetworkOverheadFilter's pod (selector and AppGroup label: makePod's first and fourth arguments), node under test and verdict (:1055-1276, 8 cases: the "Satisfied: s Violated: v" message or nil)"""
    import re
    import network as N
    src = (REF / "pkg/networkaware/networkoverhead/networkoverhead_test.go").read_text()
    checked = 0
    go = parse_literal_after(src[src.index("func TestNetworkOverheadScore"):], "tests := ")
    assert len(go) == len(N.SCORE_CASES)
    for t, c in zip(go, N.SCORE_CASES):
        idx = lambda e: e["Name"].args[0].args[1]  # nodes[i].Name
        for key, mine in (("wantedScoresBefore", c["before"]), ("wantedScoresAfter", c["after"])):
            assert [idx(e) for e in t[key]] == list(range(len(mine))) and [e["Score"] for e in t[key]] == mine, (t["name"], key)
        assert t["pod"].args[0] == c["selector"] and t["pod"].args[3] == c["appgroup"] and c["name"].split(",")[0] in t["name"], t["name"]
        checked += 1
    go = parse_literal_after(src[src.index("func TestNetworkOverheadFilter"):], "tests := ")
    assert len(go) == len(N.FILTER_CASES)
    for t, c in zip(go, N.FILTER_CASES):
        st = t["wantStatus"]
        want = None
        if isinstance(st, Call):
            m = re.search(r"Satisfied: (\d+) Violated: (\d+)", st.args[1])
            want = (int(m.group(1)), int(m.group(2)))
        assert (t["pod"].args[0], t["pod"].args[3], t["nodeToFilter"].args[1], want) == (c["selector"], c["appgroup"], c["node"], c["want"]), (t["name"], c)
        checked += 1
    return checked


def check_nrt_helpers() -> int:
    """nrt_helpers.py: RESOURCE_CLASSES (numaresources_test.go:29-115, both tables), ONLY_NON_NUMA (pluginhelpers_test.go:53-96),
    CONFIG_FROM_ATTRIBUTES / CONFIG_FROM_POLICIES / CONFIG_FROM_NRT (nodeconfig/topologymanager_test.go:256-607: every case of the Go tables, by line;
    the Go test compares the partial config the function returns, the hand-typed rows hold it applied on the defaults
    (none, container, 8) — topologymanager.go:47-53)"""
    import nrt_helpers as H
    names = {"corev1.ResourceCPU": "cpu", "corev1.ResourceMemory": "memory", "corev1.ResourceStorage": "storage",
             "corev1.ResourceEphemeralStorage": "ephemeral-storage"}
    rname = lambda v: names[v.name] if isinstance(v, Ident) else (v.args[0] if isinstance(v, Call) else names.get(v, v))
    checked = 0
    src = (REF / "pkg/noderesourcetopology/numaresources_test.go").read_text()
    got = {}
    for col, fn in enumerate(("TestIsHostLevelResource", "TestIsNUMAAffineResource")):
        for t in parse_literal_after(src[src.index("func " + fn):], "testCases := "):
            got.setdefault(rname(t["resource"]), [None, None])[col] = _num(t["expected"])
            checked += 1
    assert {k: tuple(v) for k, v in got.items()} == H.RESOURCE_CLASSES, got
    src = (REF / "pkg/noderesourcetopology/pluginhelpers_test.go").read_text()
    p = src.index("func TestOnlyNonNUMAResources")
    go = parse_literal_after(src[p:], "testCases := ")
    assert len(go) == len(H.ONLY_NON_NUMA)
    for t, (line, resources, expected) in zip(go, H.ONLY_NON_NUMA):
        want = {rname(k): v.args[0] for k, v in t["resources"].items()}
        assert want == resources and _num(t["expected"]) == expected and abs(line_of(src, '"' + t["description"] + '"', p) - line) <= 2, (t, line)
        checked += 1
    src = (REF / "pkg/noderesourcetopology/nodeconfig/topologymanager_test.go").read_text()
    policy = {"kubeletconfig.RestrictedTopologyManagerPolicy": "restricted", "kubeletconfig.SingleNumaNodeTopologyManagerPolicy": "single-numa-node",
              "kubeletconfig.BestEffortTopologyManagerPolicy": "best-effort", "kubeletconfig.NoneTopologyManagerPolicy": "none"}
    scope = {"kubeletconfig.PodTopologyManagerScope": "pod", "kubeletconfig.ContainerTopologyManagerScope": "container"}

    def applied(exp):
        exp = exp or {}
        mx = exp.get("MaxNUMANodes", 8)
        return (policy[exp["Policy"].name] if "Policy" in exp else "none", scope[exp["Scope"].name] if "Scope" in exp else "container",
                1024 if isinstance(mx, Ident) else mx)

    for fn, key, table in (("TestConfigFromAttributes", "attrs", H.CONFIG_FROM_ATTRIBUTES), ("TestConfigFromPolicies", "policies", H.CONFIG_FROM_POLICIES)):
        p = src.index("func " + fn)
        by_line = {row[0]: row for row in table}
        seen = 0
        for t in parse_literal_after(src[p:], "tests := "):
            line = line_of(src, '"' + t["name"] + '"', p)
            arg = t[key]
            if isinstance(arg, Ident):  # nil: the same case as the empty list that follows it
                assert arg.name == "nil" and applied(t["expected"]) == ("none", "container", 8)
                continue
            if key == "attrs":
                arg = {a["Name"]: a["Value"] for a in arg}
            else:
                arg = [a.args[0].name.split(".")[-1] if isinstance(a, Call) else a for a in arg]
            row = by_line[line]
            assert row[1] == arg and tuple(row[2]) == applied(t["expected"]), (fn, line, row, arg, applied(t["expected"]))
            seen += 1
        assert seen == len(table), (fn, seen, len(table))
        checked += seen
    # TestConfigFromNRT (:500-607): TopologyPolicies + Attributes of one NRT object -> the complete config (defaults: TopologyManagerDefaults)
    p = src.index("func TestConfigFromNRT")
    go = parse_literal_after(src[p:], "tests := ")
    assert len(go) == len(H.CONFIG_FROM_NRT)
    for t, (line, policies, attrs, want) in zip(go, H.CONFIG_FROM_NRT):
        nrt = t["nrt"] or {}
        got_pol = [a.args[0].name.split(".")[-1] for a in nrt.get("TopologyPolicies", [])]
        got_attr = {a["Name"]: a["Value"] for a in nrt.get("Attributes", [])}
        exp = t["expected"]
        exp_t = ("none", "container", 8) if isinstance(exp, Call) else (policy[exp["Policy"].name], scope[exp["Scope"].name], 8 if exp["MaxNUMANodes"].name == "DefaultMaxNUMANodes" else None)
        assert (got_pol, got_attr, exp_t) == (policies, attrs, tuple(want)) and line_of(src, '"' + t["name"] + '"', p) == line, (t["name"], got_pol, got_attr, exp_t)
        checked += 1
    return checked


def check_nrt_helpers_pods() -> int:
    """nrt_helpers.py: EFFECTIVE_REQUEST (pkg/util/resource_test.go:34-149, 8 cases: container, init-container and overhead
    requests as makeResourceList(cpu, mem) arguments, the expected sum), INCLUDE_NON_NATIVE (resourcerequests/exclusive_test.go
    coreTestCases :174-437, 10 cases: each container list's requests, sidecars = init containers with RestartPolicy Always,
    expectedNonNative) and MIN_DISTANCE (least_numa_test.go:758-920: the cost maps, subset size = the combinations' length, expected)"""
    import nrt_helpers as H
    checked = 0
    src = (REF / "pkg/util/resource_test.go").read_text()
    p = src.index("func TestGetPodEffectiveRequest")
    pair = lambda c: tuple(c.args) if isinstance(c, Call) else None
    go = parse_literal_after(src[p:], "tests := ")
    assert len(go) == len(H.EFFECTIVE_REQUEST)
    for t, (line, app, init, overhead, want) in zip(go, H.EFFECTIVE_REQUEST):
        lst = lambda v: [] if isinstance(v, Ident) or v is None else [pair(c) for c in v]
        got = (lst(t.get("containerRequest")), lst(t.get("initContainerRequest")), pair(t.get("podOverheadRequest")), pair(t["want"]))
        assert got == (app, init, overhead, want) and abs(line_of(src, '"' + t["name"] + '"', p) - line) <= 2, (t["name"], got, line)
        checked += 1
    src = (REF / "pkg/noderesourcetopology/resourcerequests/exclusive_test.go").read_text()
    p = src.index("func coreTestCases")
    names = {"corev1.ResourceCPU": "cpu", "corev1.ResourceMemory": "memory"}

    def requests(c):
        rl = c["Resources"].get("Requests") or {}
        lim = c["Resources"].get("Limits") or {}
        out = {names.get(k, k): v.args[0] for k, v in rl.items()}
        assert {names.get(k, k): v.args[0] for k, v in lim.items()} in (out, {}) or not out, c  # limits equal the requests in every case
        return out or {names.get(k, k): v.args[0] for k, v in lim.items()}

    go = parse_literal_after(src[p:], "return ")
    assert len(go) == len(H.INCLUDE_NON_NATIVE)
    for t, (line, name, app, init, sidecar, want) in zip(go, H.INCLUDE_NON_NATIVE):
        spec = (t["pod"].get("Spec") or {}) if isinstance(t["pod"], dict) else {}
        inits = spec.get("InitContainers") or []
        is_sidecar = lambda c: "RestartPolicy" in c
        got = ([requests(c) for c in spec.get("Containers") or []], [requests(c) for c in inits if not is_sidecar(c)],
               [requests(c) for c in inits if is_sidecar(c)], _num(t["expectedNonNative"]))
        assert t["name"] == name and got == (app, init, sidecar, want) and abs(line_of(src, '"' + name + '"', p) - line) <= 2, (name, got)
        checked += 1
    src = (REF / "pkg/noderesourcetopology/least_numa_test.go").read_text()
    p = src.index("func TestMinDistance")
    costs = {z["NUMAID"]: dict(z["Costs"]) for z in parse_literal_after(src[p:], "numaNodes := ")}
    assert costs == H.MIN_DISTANCE_COSTS, costs
    go = parse_literal_after(src[p:], "tcases := ")
    assert len(go) == len(H.MIN_DISTANCE)
    for t, (line, with_costs, size, want) in zip(go, H.MIN_DISTANCE):
        assert {len(c) for c in t["combinations"]} == {size} and (t["numaNodes"].name == "numaNodes") == with_costs and float(t["expected"]) == want, t
        assert abs(line_of(src, '"' + t["description"] + '"', p) - line) <= 2, (t["description"], line)
        checked += 1
    return checked


def check_nrt_helpers_numa_lists() -> int:
    """nrt_helpers.py: SUBTRACT_NUMA (numaresources_test.go:117-373, TestSubtractResourcesFromNUMANodeList, 9 cases: the NUMA node list,
    numaID, QoS, the container's resources, the expected list or an error) and SUBTRACT_NUMAS (:375-462, TestSubstractNUMA, 2 cases)"""
    import nrt_helpers as H
    names = {"corev1.ResourceCPU": "cpu", "corev1.ResourceMemory": "memory", "corev1.ResourceEphemeralStorage": "ephemeral-storage", "corev1.ResourceStorage": "storage"}
    qos = {"corev1.PodQOSGuaranteed": "Guaranteed", "corev1.PodQOSBurstable": "Burstable", "corev1.PodQOSBestEffort": "BestEffort"}

    def qty(v):  # mustParseQuantity(t, "8") / resource.MustParse("10Gi") / resource.NewQuantity(8, DecimalSI)
        a = v.args[-1] if v.fn == "mustParseQuantity" else v.args[0]
        return str(a)

    def rl(d):
        return {names.get(k, k.args[0] if isinstance(k, Call) else k): qty(v) for k, v in (d or {}).items()}

    def zones(lst):
        return [(z.get("NUMAID", 0), rl(z.get("Resources"))) for z in lst]

    src = (REF / "pkg/noderesourcetopology/numaresources_test.go").read_text()
    checked = 0
    p = src.index("func TestSubtractResourcesFromNUMANodeList")
    go = parse_literal_after(src[p:], "testCases := ")
    assert len(go) == len(H.SUBTRACT_NUMA)
    for t, c in zip(go, H.SUBTRACT_NUMA):
        expected = None if "expectedError" in t else zones(t["expected"])
        got = (t["name"], zones(t["nodes"]), t.get("numaID", 0), qos[t["qos"].name], rl(t.get("containerRes")), expected)
        assert got == (c["name"], c["zones"], c["numa_id"], c["qos"], c["request"], c["expected"]), (got, c)
        assert abs(line_of(src, '"' + t["name"] + '"', p) - c["line"]) <= 2, (t["name"], c["line"])
        checked += 1
    p = src.index("func TestSubstractNUMA")
    go = parse_literal_after(src[p:], "tcases := ")
    assert len(go) == len(H.SUBTRACT_NUMAS)
    for t, c in zip(go, H.SUBTRACT_NUMAS):
        got = (t["description"], zones(t["numaNodes"]), rl(t["resources"]), list(t["nodes"]), zones(t["expected"]))
        assert got == (c["name"], c["zones"], c["request"], c["nodes"], c["expected"]), (got, c)
        assert abs(line_of(src, '"' + t["description"] + '"', p) - c["line"]) <= 2
        checked += 1
    return checked


def check_peaks() -> int:
    """peaks.py: the power model fixture (peaks_test.go:80-86) and NORMALIZE_CASES (TestPeaksNormalizeScore :426-531: the score lists
    before and after, by their line)"""
    import peaks as P
    src = (REF / "pkg/trimaran/peaks/peaks_test.go").read_text()
    model = parse_literal_after(src[src.index("var data = "):], '"node-1": ')  # (the outer map's element type is `interface{}`: the inner literal is parsed)
    assert {k: float(v) for k, v in model.items()} == P.POWER_MODEL, model
    score = {"fwk.MinNodeScore": 0, "fwk.MaxNodeScore": 100}
    p = src.index("func TestPeaksNormalizeScore")
    lists = {n: [score[e["Score"].name] for e in parse_literal_after(src[p:], n + " := ")] for n in ("nodeScoreList1", "nodeScoreList2", "nodeScoreList3")}
    go = parse_literal_after(src[p:], "tests := ")
    assert len(go) == len(P.NORMALIZE_CASES)
    for t, (line, before, after) in zip(go, P.NORMALIZE_CASES):
        at = line_of(src, '"' + t["test"] + '"', p)
        assert abs(at - line) <= 3, (t["test"], at, line)  # (the hand-typed table cites a line inside the case)
        assert lists[t["nodeScoreList"].name] == before and [score[e["Score"].name] for e in t["expected"]] == after, t["test"]
    checked = 1 + len(go)
    # TestPeaksScore (:165-423): per case the pod (by the name of its fixture), the watcher response's metrics and the expected score
    # (scoreToUse = int64(jump(0, 100) * 1e15) is computed in the test; the hand-typed table marks it inexact)
    p = src.index("func TestPeaksScore")
    typ = {"watcher.CPU": "CPU", "watcher.Memory": "Memory"}
    pods = {"testPod3": P._POD3, "testPod4": P._POD4}
    go = parse_literal_after(src[p:], "tests := ")
    assert len(go) == len(P.SCORE_CASES)
    for t, c in zip(go, P.SCORE_CASES):
        pod = t["pod"]
        if isinstance(pod, Ident):
            want_pod = pods[pod.name]
        elif pod.args[0].fn == ".Container":  # testutil2.MakePod("ns", "p").Container(MakeResourceList().CPU(1).Mem(2).Obj())
            rl = pod.args[0].args[1].args[0]
            assert rl.fn == ".Mem" and rl.args[0].fn == ".CPU" and (rl.args[0].args[1], rl.args[1]) == (1, 2)
            want_pod = P._REQ
        else:
            want_pod = {"containers": []}
        resp = t["watcherResponse"]
        if not resp:
            metrics = None
        else:
            m = resp["Data"]["NodeMetricsMap"]
            metrics = {0: [(typ[x["Type"].name], "Latest", x["Value"]) for x in m["node-1"]["Metrics"]]} if m else {}
        score = t["expected"][0]["Score"]
        want_score = P.SCORE_TO_USE if score.name == "scoreToUse" else {"fwk.MinNodeScore": 0}[score.name]
        assert (t["test"], want_pod, metrics, want_score) == (c["name"], c["pod"], c["metrics"], c["expected"]) and c["exact"] == (score.name != "scoreToUse"), t["test"]
        assert -2 <= c["line"] - line_of(src, '"' + t["test"] + '"', p) <= 12, (t["test"], c["line"])
        checked += 1
    return checked


def check_trimaran_stats() -> int:
    """trimaran.py: STATS_METRICS / STATS_EXPECT (resourcestats_test.go:36-161, TestCreateResourceStats): the package-level `metrics`
    list (type, operator, value per entry), the node's capacity, the pod request and the two expected ResourceStats"""
    import re
    import trimaran as T
    src = (REF / "pkg/trimaran/resourcestats_test.go").read_text()
    typ = {"watcher.CPU": "CPU", "watcher.Memory": "Memory"}
    op = {"watcher.Average": "AVG", "watcher.Std": "STD", "": ""}
    go = parse_literal_after(src, "metrics = ")
    got = [(typ[m["Type"].name], op[m["Operator"].name if isinstance(m["Operator"], Ident) else m["Operator"]], m["Value"]) for m in go]
    assert got == [tuple(x) for x in T.STATS_METRICS], got
    f = src[src.index("func TestCreateResourceStats"):]
    pr = parse_literal_after(f, "pr := ")
    assert (eval_const(pr["MilliCPU"]), eval_const(pr["Memory"])) == (100, 1024 * 1024)  # the golden comment's "podRequest {100m, 1Mi}"
    res = parse_literal_after(src, "nodeResources = ")
    assert {str(k): v for k, v in res.items()} == {"v1.ResourceCPU": "1000m", "v1.ResourceMemory": "1Gi"}, res
    for var, key in (("rsExpectedCPU := ", "cpu"), ("rsExpectedMem := ", "memory")):
        rs = parse_literal_after(f, var)
        want = dict(capacity=float(rs["Capacity"]), req=float(rs["Req"]), used_avg=float(rs["UsedAvg"]), used_stdev=float(rs["UsedStdev"]))
        assert T.STATS_EXPECT[key] == want, (key, want)
    # the three sub-tests: cpu -> rsExpectedCPU, metrics[3:5] only -> (nil, false), memory -> rsExpectedMem
    tests = parse_literal_after(f, "tests := ")
    assert [(t["name"], t["wantIsValid"].name) for t in tests] == [("test-cpu", "true"), ("test-missing", "false"), ("test-memory", "true")]
    return len(got) + 2 + len(tests)


def check_nrt_helpers_zones_and_over_reserve() -> int:
    """nrt_helpers.py: ONLY_NON_NUMA_ZONES (pluginhelpers_test.go:29-47, the NUMANodeList TestOnlyNonNUMAResources runs on) and the two
    OVER_RESERVE flows — straight-line tests, read off with regular expressions: cache/store_test.go:998 TestResourceStoreUpdate (zones
    by MakeTopologyResInfo(name, capacity, available), the pod's two containers, every `Available.Cmp(resource.MustParse(...))`
    expectation and the missing device on zone 0) and cache/overreserve_test.go:292 TestGetCachedNRTCopyReserve over
    cache_test.go:282 makeDefaultTestTopology()"""
    import re
    import nrt_helpers as H
    checked = 0
    src = (REF / "pkg/noderesourcetopology/pluginhelpers_test.go").read_text()
    f = src[src.index("func TestOnlyNonNUMAResources"):]
    zones = parse_literal_after(f, "numaNodes := ")
    got = []
    for z in zones:
        rl = {}
        for k, v in z["Resources"].items():
            name = {"corev1.ResourceCPU": "cpu", "corev1.ResourceMemory": "memory"}.get(str(k.name if isinstance(k, Ident) else k), k if isinstance(k, str) else None)
            assert name is not None, k
            rl[name] = str(v.args[0]) if isinstance(v, Call) else v
        got.append((z["NUMAID"], rl))
    assert got == [(i, dict(r)) for i, r in H.ONLY_NON_NUMA_ZONES], got
    checked += len(got)

    names = {"cpu": "cpu", "memory": "memory", "nicName": "vendor.com/nic", "nicResourceName": "vendor.com/nic"}
    # (the reference's device is "vendor_A.com/nic" in store_test.go and another constant in cache_test.go; the hand-typed table calls
    # both "vendor.com/nic": only identity within a case matters)
    def zones_of(text):
        out = []
        for zm in re.finditer(r'Name:\s*"node-(\d)".*?ResourceInfoList\{(.*?)\},\s*\},', text, re.S):
            rl = {names[m.group(1)]: m.group(3) for m in re.finditer(r'MakeTopologyResInfo\((\w+), "([^"]+)", "([^"]+)"\)', zm.group(2))}
            out.append((int(zm.group(1)), rl))
        return out

    st = (REF / "pkg/noderesourcetopology/cache/store_test.go").read_text()
    f = st[st.index("func TestResourceStoreUpdate"):]
    f = f[:f.index("\nfunc ", 10)]
    c = H.OVER_RESERVE[0]
    assert c["source"] == "cache/store_test.go" and c["line"] == st.count("\n", 0, st.index("func TestResourceStoreUpdate")) + 1
    assert zones_of(f[:f.index("pod := ")]) == [(i, dict(r)) for i, r in c["zones"]]
    ctrs = []
    for cm in re.finditer(r'Requests: corev1\.ResourceList\{(.*?)\}', f, re.S):
        rl = {}
        for m in re.finditer(r'(corev1\.ResourceCPU|corev1\.ResourceMemory|corev1\.ResourceName\(nicName\)):\s*resource\.MustParse\("([^"]+)"\)', cm.group(1)):
            rl[{"corev1.ResourceCPU": "cpu", "corev1.ResourceMemory": "memory"}.get(m.group(1), "vendor.com/nic")] = m.group(2)
        ctrs.append(rl)
    assert [ctrs] == c["assumed_pods"], ctrs
    expected = {}
    for m in re.finditer(r'(\w+) := findResourceInfo\(nrt\.Zones\[(\d)\]\.Resources, (\w+)\)(.*?)(?=\n\t\w+ := findResourceInfo|\Z)', f, re.S):
        var, zone, res, body = m.group(1), int(m.group(2)), names[m.group(3)], m.group(4)
        av = re.search(var + r'\.Available\.Cmp\(resource\.MustParse\("([^"]+)"\)\)', body)
        if av:
            expected.setdefault(zone, {})[res] = av.group(1)
        else:
            assert re.search(r'if ' + var + r' != nil', body), (var, "neither an availability nor an absence expectation")
    assert sorted(expected.items()) == [(i, dict(r)) for i, r in c["expected"]], expected
    checked += 1

    ov = (REF / "pkg/noderesourcetopology/cache/overreserve_test.go").read_text()
    c = H.OVER_RESERVE[1]
    assert c["source"] == "cache/overreserve_test.go" and c["line"] == ov.count("\n", 0, ov.index("func TestGetCachedNRTCopyReserve(")) + 1
    f = ov[ov.index("func TestGetCachedNRTCopyReserve("):]
    f = f[:f.index("\nfunc ", 10)]
    ct = (REF / "pkg/noderesourcetopology/cache/cache_test.go").read_text()
    topo = ct[ct.index("func makeDefaultTestTopology"):]
    topo = topo[:topo.index("\n}\n") + 3]
    assert 'Name: "node1"' in topo and 'ReserveNodeResources("node1"' in f and 'GetCachedNRTCopy(context.Background(), "node1"' in f
    assert zones_of(topo) == [(i, dict(r)) for i, r in c["zones"]], zones_of(topo)
    req = re.search(r'Requests: corev1\.ResourceList\{(.*?)\}', f, re.S).group(1)
    rl = {{"corev1.ResourceCPU": "cpu", "corev1.ResourceMemory": "memory"}[m.group(1)]: m.group(2)
          for m in re.finditer(r'(corev1\.Resource\w+):\s*resource\.MustParse\("([^"]+)"\)', req)}
    assert [[rl]] == c["assumed_pods"], rl
    want = {{"corev1.ResourceCPU": "cpu", "corev1.ResourceMemory": "memory"}[m.group(1)]: m.group(2)
            for m in re.finditer(r'case string\((corev1\.Resource\w+)\):\s*if zoneRes\.Available\.Cmp\(resource\.MustParse\("([^"]+)"\)\)', f)}
    for i, r in c["expected"]:  # every zone: cpu and memory as the switch expects them, the device untouched
        assert {k: v for k, v in r.items() if k != "vendor.com/nic"} == want and r["vendor.com/nic"] == dict(c["zones"])[i]["vendor.com/nic"], (i, r, want)
    checked += 1
    return checked


def _run_pod_statements(body: str):
    """A reader for the straight-line fixture code of resourcestats_test.go's two tests — `var x int64 = n`, `x := []int64{...}`,
    reassignments, `p := getPodWithContainersAndOverhead(...)`, `p = getPodWithLimits(p, ...)` — evaluated with lroc.pod_with's
    restatement of the two builders (:612-648).  Yields ("res", pod) at every `GetResourceLimits(p)` assignment and ("want",
    cpu, mem) at every `resExpected` literal, in source order; returns the final environment as well."""
    import re
    import lroc as L
    env, events = {}, []

    def val(tok):
        tok = tok.strip()
        return env[tok] if tok in env else eval_const_text(tok)

    def eval_const_text(t):
        assert re.fullmatch(r"[\d\s*+]+", t), t
        return eval(t)  # digits, * and + only

    for line in body.split("\n"):
        line = line.strip()
        m = re.fullmatch(r"(?:var )?(\w+)(?: int64)? :?= ([\d\s*+]+)", line)
        if m:
            env[m.group(1)] = eval_const_text(m.group(2))
            continue
        m = re.fullmatch(r"(\w+) :?= \[\]int64\{([^}]*)\}", line)
        if m:
            env[m.group(1)] = [eval_const_text(x) for x in m.group(2).split(",") if x.strip()]
            continue
        m = re.fullmatch(r"(\w+) :?= getPodWithContainersAndOverhead\(([^)]*)\)", line)
        if m:
            a = [val(x) for x in m.group(2).split(",")]
            assert len(a[3]) == len(a[4])
            env[m.group(1)] = dict(ovhd=a[0], init_req=(a[1], a[2]), cont_req=list(zip(a[3], a[4])), init_lim=None, cont_lim=None)
            continue
        m = re.fullmatch(r"(\w+) = getPodWithLimits\((\w+), ([^)]*)\)", line)
        if m:
            a = [val(x) for x in m.group(3).split(",")]
            src = dict(env[m.group(2)])
            if len(a[2]) == len(a[3]) == len(src["cont_req"]):  # :634-636: otherwise the pod comes back unchanged
                src["init_lim"], src["cont_lim"] = (a[0], a[1]), list(zip(a[2], a[3]))
            env[m.group(1)] = src
            continue
        m = re.fullmatch(r"(\w+) :?= GetResourceLimits\((\w+)\)", line)
        if m:
            events.append(("res", env[m.group(2)]))
            continue
    for m in re.finditer(r"resExpected :?= &framework\.Resource\{\s*MilliCPU:\s*([\d\s*]+),\s*Memory:\s*([\d\s*]+),\s*\}", body):
        events.append(("want", m.start(), eval_const_text(m.group(1)), eval_const_text(m.group(2))))
    to_pod = lambda d: L.pod_with(d["ovhd"], d["init_req"], d["cont_req"], d["init_lim"], d["cont_lim"])
    return env, events, to_pod


def check_lroc_resource_tables() -> int:
    """lroc.py: RESOURCE_LIMITS (resourcestats_test.go:203-257, TestGetResourceLimits: four pods built by two helper calls each and
    the expected (milliCPU, memory) after each), NODE_REQUESTS_LIMITS (:374-603, TestGetNodeRequestsAndLimits: the four pods, which
    of them sit on the node, the two nodes, and `want` — literal products for test-0 / test-1; for test-2 / test-3 the reference
    writes min / max expressions over GetResourceRequested / GetResourceLimits and the node's capacity, evaluated here with the
    hand-typed pod's own sums) and SCORE_CASES (lowriskovercommitment_test.go:138-243: one case)"""
    import re
    import lroc as L
    src = (REF / "pkg/trimaran/resourcestats_test.go").read_text()
    checked = 0
    f = src[src.index("func TestGetResourceLimits"):]
    f = f[:f.index("\nfunc ", 10)]
    env, events, to_pod = _run_pod_statements(f)
    pods = [to_pod(d) for k, d in [(e[0], e[1]) for e in events if e[0] == "res"]]
    # GetResourceLimits(pod0) is assigned to `res0`: same statement shape
    wants = [(e[2], e[3]) for e in events if e[0] == "want"]
    assert len(pods) == len(wants) == len(L.RESOURCE_LIMITS) == 4, (len(pods), len(wants))
    for (pod, cpu, mem), got_pod, (wc, wm) in zip(L.RESOURCE_LIMITS, pods, wants):
        assert pod == got_pod and (cpu, mem) == (wc, wm), (cpu, mem, wc, wm)
        checked += 1

    f = src[src.index("func TestGetNodeRequestsAndLimits"):]
    f = f[:f.index("\nfunc ", 10)]
    env, _, to_pod = _run_pod_statements(f[:f.index("tests := ")])
    # the statements reassign contCPULimit / contMemLimit between the pods: pod..pod2 share one shape, pod3 and pod4 differ
    want_pods = {"pod": L.POD, "pod1": L.POD, "pod2": L.POD, "pod3": L.POD3, "pod4": L.POD4}
    for name, want in want_pods.items():
        assert to_pod(env[name]) == want, name
    nodes = [m.group(1) for m in re.finditer(r"v1\.ResourceCPU:\s*\"(\w+)\"", f[:f.index("tests := ")])]
    mems = [m.group(1) for m in re.finditer(r"v1\.ResourceMemory:\s*\"(\w+)\"", f[:f.index("tests := ")])]
    assert ({"cpu": nodes[0], "memory": mems[0]}, {"cpu": nodes[1], "memory": mems[1]}) == (L.TEST_NODE, L.LOW_NODE)
    node_of = {"testNode": L.TEST_NODE, "testNodeWithLowNodeCapacity": L.LOW_NODE}
    info_of = {"podInfo1": L.POD, "podInfo2": L.POD, "podInfo3": L.POD3}

    def sums(pod):  # GetResourceRequested / GetResourceLimits of a pod_with() pod: max(sum of containers, init container) + overhead
        q = lambda d, k: int(str(d.get(k, 0)).rstrip("m")) if k == "cpu" else int(d.get(k, 0))
        out = {}
        for what in ("requests", "limits"):
            for k in ("cpu", "memory"):
                tot = sum(q(c[what], k) for c in pod["containers"])
                tot = max(tot, max((q(c[what], k) for c in pod["init_containers"]), default=0))
                out[(what, k)] = tot + (q(pod["overhead"], k) if k == "cpu" else 0)
        return out

    tests = parse_literal_after(f, "tests := ")
    assert len(tests) == len(L.NODE_REQUESTS_LIMITS)
    for t, c in zip(tests, L.NODE_REQUESTS_LIMITS):
        a = t["args"]
        assert t["name"] == c["name"] and [info_of[x.name] for x in a["podsOnNode"]] == c["on_node"] and node_of[a["node"].name] == c["node"], t["name"]
        assert want_pods[a["pod"].name] == c["pod"], t["name"]
        cap_cpu, cap_mem = int(c["node"]["cpu"].rstrip("m")), {"6Ki": 6 * 1024}[c["node"]["memory"]]

        def ev(v):
            if isinstance(v, int):
                return v
            if isinstance(v, Ident):
                return {"capCpu": int(L.LOW_NODE["cpu"].rstrip("m")), "capMem": 6 * 1024}[v.name]
            if isinstance(v, Call) and v.fn in ("op*", "op+"):
                return eval_const(v)
            if isinstance(v, Call) and v.fn in ("min", "max"):
                vals = [ev(x) for x in v.args]
                return min(vals) if v.fn == "min" else max(vals)
            if isinstance(v, Call) and v.fn == "select":  # GetResourceRequested(pod4).MilliCPU
                inner, field = v.args
                s_ = sums(want_pods[inner.args[0].name])
                return s_[({"GetResourceRequested": "requests", "GetResourceLimits": "limits"}[inner.fn], "cpu" if field == "MilliCPU" else "memory")]
            raise ValueError(repr(v))

        w = t["want"]
        got = dict(req_cpu=ev(w["NodeRequest"]["MilliCPU"]), req_mem=ev(w["NodeRequest"]["Memory"]), lim_cpu=ev(w["NodeLimit"]["MilliCPU"]),
                   lim_mem=ev(w["NodeLimit"]["Memory"]), req_minus_pod_cpu=ev(w["NodeRequestMinusPod"]["MilliCPU"]),
                   req_minus_pod_mem=ev(w["NodeRequestMinusPod"]["Memory"]), lim_minus_pod_cpu=ev(w["NodeLimitMinusPod"]["MilliCPU"]),
                   lim_minus_pod_mem=ev(w["NodeLimitMinusPod"]["Memory"]), cap_cpu=ev(w["Nodecapacity"]["MilliCPU"]), cap_mem=ev(w["Nodecapacity"]["Memory"]))
        assert (got["cap_cpu"], got["cap_mem"]) == (cap_cpu, cap_mem), t["name"]
        assert got == c["want"], (t["name"], got, c["want"])
        checked += 1

    src = (REF / "pkg/trimaran/lowriskovercommitment/lowriskovercommitment_test.go").read_text()
    p0 = src.index("func TestLowRiskOverCommitment_Score")
    tests = parse_literal_after(src[p0:], "tests := ")
    assert len(tests) == len(L.SCORE_CASES) == 1
    t, c = tests[0], L.SCORE_CASES[0]
    m = t["watcherResponse"]["Data"]["NodeMetricsMap"]["node-1"]["Metrics"]
    typ = {"watcher.CPU": "CPU", "watcher.Memory": "Memory"}
    op = {"watcher.Average": "AVG", "watcher.Std": "STD", "watcher.Latest": "Latest"}
    assert {0: [(typ[x["Type"].name], op[x["Operator"].name], x["Value"]) for x in m]} == c["metrics"], m
    assert [eval_const(e["Score"], {"fwk.MinNodeScore": 0, "fwk.MaxNodeScore": 100}) for e in t["expected"]] == c["expected"]
    assert abs(c["line"] - line_of(src, '"' + t["test"] + '"', p0)) <= 10
    checked += 1
    return checked


def _nrt_zones(lit):
    """Zones of a NodeResourceTopology literal -> the hand-typed form: [{name, type, resources: [(name, capacity, allocatable, available)]}]"""
    def q(v):
        assert isinstance(v, Call) and v.fn == "resource.MustParse", v
        return v.args[0]
    return [{"name": z["Name"], "type": z.get("Type", "Node"),
             "resources": [(r["Name"], q(r["Capacity"]), q(r["Allocatable"]), q(r["Available"])) for r in z["Resources"]]} for z in lit["Zones"]]


def check_nrt_preemption() -> int:
    """nrt_preemption.py against preemption_test.go: the two fixtures (getTestNRT, getTestEncodedInfo10Containers' affinity list) and every
    case of TestGetNRTPostPodsEviction — victims (namespace, name, QOS class, containers' requests and limits), which placement record the case
    passes, the expected error text, the expected zone table (getTestNRT() again wherever the case expects an error) and the case's line."""
    import nrt_preemption as N
    src = (REF / "pkg/noderesourcetopology/preemption/preemption_test.go").read_text()
    checked = 0
    fix = parse_literal_after(src[src.index("func getTestNRT"):], "return ")
    zones = _nrt_zones(fix)
    for z in zones:
        z["type"] = "Node"  # (the fixture leaves Type empty; the hand-typed table fills the CRD's value for the flattener — nothing reads it on this path)
    assert zones == N.TEST_NRT["zones"], zones
    checked += 1
    aff = parse_literal_after(src[src.index("func getTestEncodedInfo10Containers"):], "affinities := ")
    got = {(a["ID"]["Namespace"], a["ID"]["PodName"], a["ID"]["ContainerName"]): a["NUMANode"] for a in aff}
    assert got == N.PLACEMENT and len(aff) == len(got) == 10, got
    checked += 1
    qos = {"corev1.PodQOSGuaranteed": N.G, "corev1.PodQOSBurstable": N.BU, "corev1.PodQOSBestEffort": N.BE}
    p0 = src.index("func TestGetNRTPostPodsEviction")
    tests = parse_literal_after(src[p0:], "testcases := ")
    assert len(tests) == len(N.CASES), (len(tests), len(N.CASES))

    def rl(m):
        return {k: v.args[0] for k, v in (m or {}).items()}

    def pod_qos(v, ctrs):
        if "Status" in v:
            return qos[v["Status"]["QOSClass"].name]
        # no Status.QOSClass: v1qos.GetPodQOS (preemption.go:71) computes the class — a pod without requests or limits is BestEffort
        assert not any(k["requests"] or k["limits"] for k in ctrs), v
        return N.BE

    for t, c in zip(tests, N.CASES):
        assert t["name"].split(",")[0] == c["name"] or t["name"].startswith(c["name"]), (t["name"], c["name"])
        assert c["line"] == line_of(src, '"' + t["name"] + '"', p0), (t["name"], c["line"], line_of(src, '"' + t["name"] + '"', p0))
        victims = []
        for v in t["victims"]:
            ctrs = [N.ctr(k["Name"], rl(k.get("Resources", {}).get("Requests")), rl(k.get("Resources", {}).get("Limits"))) for k in v.get("Spec", {}).get("Containers", [])]
            victims.append(dict(ns=v["ObjectMeta"].get("Namespace", ""), name=v["ObjectMeta"]["Name"], qos=pod_qos(v, ctrs), containers=ctrs))
        assert victims == c["victims"], (t["name"], victims, c["victims"])
        info = t.get("numaPlacementInfo")
        if info is None:
            assert c["placement"] is None, t["name"]
        elif isinstance(info, Call) and info.fn == "getTestEncodedInfo10Containers":
            assert c["placement"] is N.PLACEMENT, t["name"]
        else:  # numaplacement.NewEncodedInfo(): a record with no containers
            assert isinstance(info, Call) and info.fn == "numaplacement.NewEncodedInfo" and not info.args and c["placement"] == {}, (t["name"], info)
        assert t.get("expectedError", "") == c["error"], (t["name"], t.get("expectedError"))
        assert c["error"] in N.ERROR_CODES, c["error"]
        want = t["expectedUpdatedNRT"]
        if isinstance(want, Call):
            assert want.fn == "getTestNRT" and "expected" not in c and c["error"], t["name"]
        else:
            wz = _nrt_zones(want)
            for z in wz:
                z["type"] = "Node"
            assert not c["error"] and wz == c["expected"]["zones"], (t["name"], wz)
        assert isinstance(t["nrt"], Call) and t["nrt"].fn == "getTestNRT", t["name"]
        checked += 1
    # the messages ERROR_CODES numbers are the ones preemption.go raises (the seventh, "NRT not found", has no case in the table)
    go = (REF / "pkg/noderesourcetopology/preemption/preemption.go").read_text()
    for msg in N.ERROR_CODES:
        assert not msg or '"' + msg + '"' in go, msg
    checked += 1
    return checked


def check_nrt_preemption_flow() -> int:
    """nrt_preemption_flow.py against filter_preemption_test.go: makePreemptionNRT's zones and policy, makeGuaranteedPod's shape, the victim /
    preemptor / placement constants of TestFilter_PreemptionFlow, and per sub-test (read as statements: they are straight-line code) the
    preemption mode, whether the cache holds a placement record, the victims on the cycle state, the preemptor pod, the expected status and
    the expected over-reserve marking."""
    import re
    import nrt_preemption_flow as F
    src = (REF / "pkg/noderesourcetopology/filter_preemption_test.go").read_text()
    checked = 0
    nrt = parse_literal_after(src[src.index("func makePreemptionNRT"):], "return ")
    names = {"cpu": "cpu", "memory": "memory"}  # the package's test constants (filter_test.go: cpu = string(corev1.ResourceCPU), ...)
    zones = []
    for z in nrt["Zones"]:
        res = []
        for r in z["Resources"]:
            assert isinstance(r, Call) and r.fn == "makeTopologyResInfoWithAllocatable", r
            nm, cap, av = r.args
            res.append((names[nm.name], cap, cap, av))  # (:81-88: Capacity = Allocatable = the second argument)
        zones.append({"name": z["Name"], "type": z["Type"], "resources": res})
    assert zones == F.NRT["zones"], zones
    pol = nrt["TopologyPolicies"]
    assert len(pol) == 1 and pol[0].args[0].name == "topologyv1alpha2." + F.NRT["policies"][0], pol
    checked += 1
    # makeNodeFromNRT: capacity = allocatable = the zones' capacities summed (makeResourceListFromZones)
    gi = lambda s_: int(s_[:-2]) if s_.endswith("Gi") else int(s_)
    assert F.NODE == {"cpu": str(sum(gi(z["resources"][0][1]) for z in zones)), "memory": "%dGi" % sum(gi(z["resources"][1][1]) for z in zones)}
    checked += 1
    p0 = src.index("func TestFilter_PreemptionFlow")
    p1 = src.index("\nfunc ", p0 + 10)
    body = src[p0:p1]
    cname = F.VICTIM["containers"][0]["name"]

    def pod(call):
        m = re.fullmatch(r'makeGuaranteedPod\("([^"]*)", "([^"]*)", containerName, (\d+), "([^"]*)"\)', call.strip())
        assert m, call
        r = {"cpu": m.group(3), "memory": m.group(4)}
        return dict(ns=m.group(1), name=m.group(2), containers=[dict(name=cname, requests=r, limits=r)])

    head = body[:body.index("t.Run(")]
    top = {m.group(1): pod(m.group(2)) for m in re.finditer(r"(\w+) := (makeGuaranteedPod\([^\n]*\))", head)}
    assert top["victim"] == F.VICTIM and top["preemptor"] == F.guaranteed("default", "preemptor", 4, "1Gi"), top
    assert "numaPlacement := makeEncodedInfoForPod(victim, 0)" in head
    assert F.PLACEMENT == {(F.VICTIM["ns"], F.VICTIM["name"], cname): 0}
    checked += 1
    subs = list(re.finditer(r't\.Run\("([^"]*)", func\(t \*testing\.T\) \{\n(.*?)\n\t\}\)', body, re.S))
    assert len(subs) == len(F.CASES), (len(subs), len(F.CASES))
    for m, c in zip(subs, F.CASES):
        name, text = m.group(1), m.group(2)
        # the hand-typed names abbreviate; every word of the abbreviation must come from the sub-test's name, and the line ties them
        assert c["line"] == src.count("\n", 0, p0 + m.start()) + 1, (name, c["line"])
        assert all(w.strip("(),:") in name for w in c["name"].replace(":", " ").split()), (name, c["name"])
        local = {k: pod(v) for k, v in re.findall(r"(\w+) := (makeGuaranteedPod\([^\n]*\))", text)}
        pods = {**top, **local}
        cache = re.search(r"cache := &fakeFilterCache\{([^}]*)\}", text).group(1)
        assert ("numaPlacement: numaPlacement" in cache) == (c["placement"] is not None), name
        mode = re.search(r"preemptionMode: apiconfig\.(\w+)\}", text).group(1)
        assert {"PreemptionEnabled": True, "PreemptionDisabled": False}[mode] == c["enabled"], name
        cs = re.search(r"cycleState := cycleStateWithVictims\(t, ([^)]*)\)", text)
        victims = [pods[v.strip()] for v in cs.group(1).split(",")] if cs else []
        assert victims == c["victims"], (name, victims)
        f = re.search(r"tm\.Filter\(context\.Background\(\), ([^,]*), (\w+), nodeInfo\)", text)
        assert (f.group(1) == "cycleState") == bool(cs) and pods[f.group(2)] == c["preemptor"], name
        st = re.search(r'quasiEqualStatus\(status, (nil|fwk\.NewStatus\(fwk\.Unschedulable, "([^"]*)"\))\)', text)
        assert (None if st.group(1) == "nil" else st.group(2)) == c["want"], (name, st.group(0))
        if "len(cache.maybeOverReserved) != 1" in text:
            assert c["over_reserved"] is True, name
        elif "len(cache.maybeOverReserved) != 0" in text:
            assert c["over_reserved"] is False, name
        else:
            assert c["over_reserved"] is None, name
        checked += 1
    return checked


def _watcher_metrics(src: str):
    """the `metrics := watcher.WatcherMetrics{...}` literal of an integration test -> (Window.End or 0, {node index: [(type, operator, value)]}) with nodes
    numbered by their name's suffix (node-1 -> 0), the hand-typed form"""
    m = parse_literal_after(src, "metrics := ")
    typ = {"watcher.CPU": "CPU", "watcher.Memory": "Memory"}
    op = {"watcher.Average": "AVG", "watcher.Std": "STD", "watcher.Latest": "Latest"}
    out = {}
    for name, nm in m["Data"]["NodeMetricsMap"].items():
        out[int(name.split("-")[1]) - 1] = [(typ[x["Type"].name], op[x["Operator"].name], float(x["Value"])) for x in nm["Metrics"]]
    return (m.get("Window") or {}).get("End", 0), out


def check_integration() -> int:
    """integration.py against test/integration/{targetloadpacking,loadVariationRiskBalancing,allocatable,lowriskovercommitment,peaks}_test.go.  The Go tests
    are straight-line code around one literal each: the literal (watcher metrics, the Allocatable case table, the power-model map) is read with goparse,
    the scalars around it (node names, NewQuantity / NewMilliQuantity arguments, the per-pod cpu lists, the expected placements) with regular expressions."""
    import re
    import integration as I
    checked = 0
    RL = {"v1.ResourcePods": "pods", "v1.ResourceCPU": "cpu", "v1.ResourceMemory": "memory", "corev1.ResourceCPU": "cpu", "corev1.ResourceMemory": "memory"}

    def strs(src, var):
        return re.search(var + r" := \[\]string\{([^}]*)\}", src).group(1).replace('"', "").replace(" ", "").split(",")

    def ints(src, var):
        return [int(x) for x in re.search(var + r" := \[\]int64\{([^}]*)\}", src).group(1).replace(" ", "").split(",")]

    def qty_lists(block):
        """resource lists written as `v1.ResourceX: *resource.NewQuantity(n, ...)` lines: one dict per `ResourceList{` in the block"""
        out = []
        for body in re.findall(r"ResourceList\{\n(.*?)\n\t*\}", block, re.S):
            out.append({RL[k]: v for k, v in re.findall(r"(\w+\.Resource\w+):\s+\*resource\.NewQuantity\((\d+), resource\.DecimalSI\)", body)})
        return out

    def expected_of(src, names):
        e = re.search(r"expected := \[2\]string\{([^}]*)\}", src).group(1).replace(" ", "").split(",")
        return [x.strip('"') if x.startswith('"') else names[int(re.fullmatch(r"nodeNames\[(\d)\]", x).group(1))] for x in e]

    def line(src, fn):
        return line_of(src, "func " + fn)

    # --- TargetLoadPacking
    src = (REF / "test/integration/targetloadpacking_test.go").read_text()
    c = I.TLP
    end, met = _watcher_metrics(src)
    names = strs(src, "nodeNames")
    assert (end, met) == (c["window_end"], c["metrics"]) and names == [n["name"] for n in c["nodes"]], (end, met)
    alloc, cap = qty_lists(src[src.index("node.Status.Allocatable"):src.index("var newPods")])
    assert all(n["allocatable"] == alloc and n["capacity"] == cap for n in c["nodes"]), (alloc, cap)
    mem = re.search(r"v1\.ResourceMemory: \*resource\.NewQuantity\((\d+), resource\.DecimalSI\),\n\t\t\t\},\n\t\t\}\n\t\tnewPods", src).group(1)
    assert c["pods"] == [{"cpu": f"{m_}m", "memory": mem} for m_ in ints(src, "containerCPU")] and len(c["pods"]) == len(strs(src, "podNames"))
    assert expected_of(src, names) == c["expected"] and abs(line(src, "TestTargetNodePackingPlugin") - c["line"]) <= 10
    assert "TargetUtilization:         cfgv1.DefaultTargetUtilizationPercent" in src and "DefaultRequestsMultiplier: cfgv1.DefaultRequestsMultiplier" in src
    checked += 1

    # --- LoadVariationRiskBalancing
    src = (REF / "test/integration/loadVariationRiskBalancing_test.go").read_text()
    c = I.LVRB
    end, met = _watcher_metrics(src)
    names = strs(src, "nodeNames")
    assert (end, met) == (c["window_end"], c["metrics"]) and names == [n["name"] for n in c["nodes"]], (end, met)
    capacity = {RL[k]: v for k, v in re.findall(r"(v1\.Resource\w+):\s+\"(\w+)\",", src[src.index("capacity := map"):src.index("for i := 0; i < len(nodeNames)")])}
    assert all(n["capacity"] == capacity and n["allocatable"] == capacity for n in c["nodes"]), capacity  # (NodeWrapper.Capacity sets both lists)
    mem = re.search(r"v1\.ResourceMemory: \*resource\.NewQuantity\((\d+), resource\.DecimalSI\),\n\t\t\t\},\n\t\t\}\n\t\tnewPods", src).group(1)
    assert c["pods"] == [{"cpu": f"{m_}m", "memory": mem} for m_ in ints(src, "containerCPU")]
    assert expected_of(src, names) == c["expected"] and abs(line(src, "TestLoadVariationRiskBalancingPlugin") - c["line"]) <= 10
    checked += 1

    # --- NodeResourcesAllocatable
    src = (REF / "test/integration/allocatable_test.go").read_text()
    c = I.ALLOCATABLE

    def rmap(var):
        body = src[src.index(var + " := map"):]
        return {RL[k]: v for k, v in re.findall(r"(v1\.Resource\w+):\s+\"(\w+)\",", body[:body.index("}")])}

    maps = {k: rmap(k) for k in ("smallNodeCapacity", "bigNodeCapacity", "smallPodReq", "bigPodReq")}
    tests = parse_literal_after(src, "testCases := ")
    assert len(tests) == 2

    def chain(v):  # st.MakeX().Name(n)...Req(m) / .Capacity(m): (name, the map variable's name)
        name = arg = None
        while isinstance(v, Call):
            if v.fn == ".Name":
                name = v.args[1]
            if v.fn in (".Req", ".Capacity"):
                arg = v.args[1].name
            v = v.args[0] if v.args and v.fn.startswith(".") else None
        return name, arg

    for t in tests:
        mode = {"schedconfig.Least": "Least", "schedconfig.Most": "Most"}[t["modeType"].name]
        assert [(n, maps[a]) for n, a in map(chain, t["pods"])] == c["pods"], t["name"]
        nodes = [(n, maps[a]) for n, a in map(chain, t["nodes"])]
        assert nodes == [(n["name"], n["capacity"]) for n in c["nodes"]] and all(n["capacity"] == n["allocatable"] for n in c["nodes"]), nodes
        want = {k: set(v.args) for k, v in t["expectedNodes"].items()}
        assert all(isinstance(v, Call) and v.fn == "sets.New" for v in t["expectedNodes"].values()) and want == c["expected"][mode], (mode, want)
        checked += 1
    w = re.search(r"\{Name: string\(v1\.ResourceMemory\), Weight: (\d+)\}", src)
    assert c["weights"] == {"memory": int(w.group(1))} and src.count("Weight:") == 2  # (the other one is the plugin's own weight in the profile)
    assert abs(line(src, "TestAllocatablePlugin") - c["line"]) <= 10

    # --- LowRiskOverCommitment
    src = (REF / "test/integration/lowriskovercommitment_test.go").read_text()
    c = I.LROC
    end, met = _watcher_metrics(src)
    names = strs(src, "nodeNames")
    assert met == c["metrics"] and names == [n["name"] for n in c["nodes"]], met
    cpu, memc = strs(src, "capCPU"), strs(src, "capMemory")
    assert [n["capacity"] for n in c["nodes"]] == [{"cpu": a, "memory": b} for a, b in zip(cpu, memc)] and all(n["capacity"] == n["allocatable"] for n in c["nodes"])
    req, lim = ints(src, "requestCPU"), ints(src, "limitCPU")
    rm, lm = re.search(r"var requestMemory int64 = (\d+)", src).group(1), re.search(r"var limitMemory int64 = (\d+)", src).group(1)
    pods = [({"cpu": f"{a}m", "memory": rm}, {"cpu": f"{b}m", "memory": lm}) for a, b in zip(req, lim)]
    sched = strs(src, "scheduledNodes")
    assert "Node(nodeNames[i])" in src and sched == names  # existing pod i sits on node i
    assert c["on_node"] == {i: [pods[i]] for i in range(len(sched))} and c["pod"] == pods[len(sched)] and len(pods) == len(sched) + 1, pods
    assert strs(src, "expectedNodes") == [c["expected"]]
    wts = parse_literal_after(src, "RiskLimitWeights: ")
    assert (float(wts["cpu"]), float(wts["memory"])) == (c["params"]["w_cpu"], c["params"]["w_mem"]) and "SmoothingWindowSize: v1.DefaultSmoothingWindowSize" in src
    d = (REF / "apis/config/v1/defaults.go").read_text()
    assert int(re.search(r"DefaultSmoothingWindowSize\s*(?:int64)?\s*=\s*(\d+)", d).group(1)) == c["params"]["smoothing_window_size"]
    assert abs(line(src, "TestLowRiskOverCommitmentPlugin") - c["line"]) <= 10
    checked += 1

    # --- Peaks
    src = (REF / "test/integration/peaks_test.go").read_text()
    c = I.PEAKS
    data = parse_literal_after(src, "data := ")
    names = strs(src, "nodeNames")
    def fl(v):  # goparse reads -x as a negation call or a negative number depending on context
        if isinstance(v, Call) and v.fn in ("op-", "neg"):
            return -float(v.args[-1])
        return float(v)
    models = [{k: fl(v) for k, v in data[n].items()} for n in names]
    assert models == c["models"], models
    end, met = _watcher_metrics(src)
    assert (end, met) == (0, c["metrics"]) and names == [n["name"] for n in c["nodes"]]
    alloc, cap = qty_lists(src[src.index("node.Status.Allocatable"):src.index("var newPods")])
    assert all(n["allocatable"] == alloc and n["capacity"] == cap for n in c["nodes"]), (alloc, cap)
    mem = re.search(r"v1\.ResourceMemory: \*resource\.NewQuantity\((\d+), resource\.DecimalSI\),\n\t\t\t\},\n\t\t\}\n\t\tnewPods", src).group(1)
    assert c["pods"] == [{"cpu": f"{m_}m", "memory": mem} for m_ in ints(src, "containerCPU")]
    assert expected_of(src, names) == c["expected"] and abs(line(src, "TestPeaksPlugin") - c["line"]) <= 10
    # `feasible` is this repository's statement of upstream's NodeResourcesFit (not in the Go file): pod-2's 1900m next to pod-1's 300m exceed node-1's 2 cpus
    cpu_m = [int(p_["cpu"].rstrip("m")) for p_ in c["pods"]]
    first = names.index(c["expected"][0])
    assert c["feasible"] == [[1] * len(names), [int(not (i == first and cpu_m[0] + cpu_m[1] > 1000 * int(cap["cpu"]))) for i in range(len(names))]]
    checked += 1
    return checked


if __name__ == "__main__":
    print("allocatable.py:", check_allocatable(), "cases agree with allocatable_test.go")
    print("trimaran.py:", check_trimaran(), "rows of COMPUTE_SCORE / MU_SIGMA agree with analysis_test.go / resourcestats_test.go")
    print("trimaran.py:", check_trimaran_score_cases(), "TLP / LVRB Score cases agree with targetloadpacking_test.go / loadvariationriskbalancing_test.go")
    print("lroc.py:", check_lroc(), "rows of the three beta tables agree with beta_test.go")
    print("lroc.py:", check_lroc_compute_risk(), "computeRisk fixtures and cases agree with lowriskovercommitment_test.go")
    print("network.py:", check_network(), "Score / Filter cases agree with networkoverhead_test.go")
    print("nrt_helpers.py:", check_nrt_helpers(), "rows (resource classes, onlyNonNUMAResources, ConfigFromAttributes / ConfigFromPolicies / ConfigFromNRT) agree with the Go tables")
    print("nrt_helpers.py:", check_nrt_helpers_pods(), "rows (GetPodEffectiveRequest, IncludeNonNative, minAvgDistanceInCombinations) agree with the Go tables")
    print("nrt_helpers.py:", check_nrt_helpers_numa_lists(), "subtract cases agree with numaresources_test.go")
    print("peaks.py:", check_peaks(), "fixtures agree with peaks_test.go")
    print("lroc.py:", check_lroc_resource_tables(), "rows (GetResourceLimits, GetNodeRequestsAndLimits, the Score case) agree with resourcestats_test.go / lowriskovercommitment_test.go")
    print("trimaran.py:", check_trimaran_stats(), "items of STATS_METRICS / STATS_EXPECT agree with resourcestats_test.go (TestCreateResourceStats)")
    print("nrt_helpers.py:", check_nrt_helpers_zones_and_over_reserve(), "items (ONLY_NON_NUMA_ZONES, the two OVER_RESERVE flows) agree with pluginhelpers_test.go / cache/*_test.go")
    print("nrt_preemption.py:", check_nrt_preemption(), "items (two fixtures, every TestGetNRTPostPodsEviction case, the error texts) agree with preemption_test.go / preemption.go")
    print("nrt_preemption_flow.py:", check_nrt_preemption_flow(), "items (NRT, node, pods, the seven sub-tests) agree with filter_preemption_test.go")
    print("integration.py:", check_integration(), "items (TLP, LVRB, Allocatable Least / Most, LROC, Peaks) agree with test/integration/*_test.go")
