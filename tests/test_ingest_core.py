"""Wire-format ingestion of v1.Node / v1.Pod: the host decoder against the independent Python builders,
on the reference's Online Boutique pods (tests/golden/pod_manifests.json), on the pods of its NRT filter tables, and on
synthetic objects; then end to end through the host flatteners.  CPU only."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

import scheduler_plugins_amd as spx
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd.ingest import NrtIngest

GOLD = Path(__file__).parent / "golden"
GROUP, SELECTOR = "appgroup.diktyo.x-k8s.io", "appgroup.diktyo.x-k8s.io.workload"


def col(struct, name, n):
    return np.ctypeslib.as_array(getattr(struct, name), (n,)).tolist() if n else []


def pods_equal(a, b):
    n = a.n_pods
    assert n == b.n_pods
    nc = a.ctr_ptr[n]
    sizes = dict(ctr_ptr=n + 1, ctr_kind=nc, req_ptr=nc + 1, lim_ptr=nc + 1, ovh_ptr=n + 1, priority=n, queue_ts=n, appgroup=n, selector=n, ns=n)
    for c, k in sizes.items():
        assert col(a, c, k) == col(b, c, k), c
    for ptr, res, qty, k in (("req_ptr", "req_res", "req_qty", nc), ("lim_ptr", "lim_res", "lim_qty", nc), ("ovh_ptr", "ovh_res", "ovh_qty", n)):
        m = getattr(a, ptr)[k]
        assert col(a, res, m) == col(b, res, m), res
        assert col(a, qty, m) == col(b, qty, m), qty


def pod_json_to_dict(p, groups, selectors, namespaces):
    """the same pod in the dict form objects.build_pod_objects takes"""
    def ctr(c, init):
        rs = c.get("resources") or {}
        return {"requests": rs.get("requests") or {}, "limits": rs.get("limits") or {}, "sidecar": init and c.get("restartPolicy") == "Always"}
    spec, meta = p.get("spec", {}), p.get("metadata", {})
    labels = meta.get("labels") or {}
    return O.pod([ctr(c, False) for c in spec.get("containers", [])], [ctr(c, True) for c in spec.get("initContainers", [])],
                 overhead=spec.get("overhead"), priority=spec.get("priority") or 0,
                 appgroup=groups.id(labels[GROUP]) if labels.get(GROUP) else -1,
                 selector=selectors.id(labels[SELECTOR]) if labels.get(SELECTOR) else -1, ns=namespaces.id(meta.get("namespace", "")))


def intern_like_decoder(res, pods_json):
    """resource ids in the order the decoder meets them: per pod overhead/containers as they appear in the document"""
    for p in pods_json:
        spec = p.get("spec", {})
        for key in spec:  # document order of the members
            if key in ("containers", "initContainers"):
                for c in spec[key]:
                    rs = c.get("resources") or {}
                    for part in rs:
                        if part in ("requests", "limits"):
                            for r in rs[part] or {}:
                                res.id(r)
            elif key == "overhead":
                for r in spec[key] or {}:
                    res.id(r)


def test_online_boutique_pods(hdr):
    doc = json.loads((GOLD / "pod_manifests.json").read_text())
    pods_json = doc["items"]
    assert len(pods_json) == 11
    res, groups, selectors, namespaces = O.Resources(), O.Interner(), O.Interner(), O.Interner()
    intern_like_decoder(res, pods_json)
    want = O.build_pod_objects(hdr, res, [pod_json_to_dict(p, groups, selectors, namespaces) for p in pods_json])
    with NrtIngest(["n0"]) as ing:
        assert ing.feed_pods(json.dumps({"kind": "PodList", "items": pods_json}).encode()) == 11
        pods_equal(ing.pod_objects().struct, want.struct)
        assert ing.name_id("appgroup", "online-boutique") == 0 and ing.name_id("selector", "adservice") == selectors.ids["adservice"]
        t = ing.pod_objects().struct
        assert sum(col(t, "req_ptr", t.ctr_ptr[11] + 1)) > 0      # the manifests do carry cpu / memory requests
        assert set(col(t, "appgroup", 11)) == {0} and len(set(col(t, "selector", 11))) == 11


def test_filter_table_pods_round_trip(hdr):
    """the pods of filter_test.go's tables (init containers, sidecars, devices, hugepages) rendered as v1.Pod JSON"""
    cases = json.loads((GOLD / "nrt_filter.json").read_text())
    dicts = [c["pod"] for g in ("cases", "pod_scope_cases", "container_scope_cases") for c in cases[g]]
    assert len(dicts) > 60

    def to_json(d, i):
        def ctr(c, init):
            out = {"name": "c", "resources": {"requests": {k: str(v) for k, v in (c.get("requests") or {}).items()},
                                              "limits": {k: str(v) for k, v in (c.get("limits") or {}).items()}}}
            if init and c.get("sidecar"):
                out["restartPolicy"] = "Always"
            return out
        spec = {"containers": [ctr(c, False) for c in d.get("containers", [])]}
        if d.get("init_containers"):
            spec["initContainers"] = [ctr(c, True) for c in d["init_containers"]]
        if d.get("overhead"):
            spec["overhead"] = {k: str(v) for k, v in d["overhead"].items()}
        return {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": f"p{i}", "namespace": "default"}, "spec": spec}

    pods_json = [to_json(d, i) for i, d in enumerate(dicts)]
    res, groups, selectors, namespaces = O.Resources(), O.Interner(), O.Interner(), O.Interner()
    intern_like_decoder(res, pods_json)
    want = O.build_pod_objects(hdr, res, [pod_json_to_dict(p, groups, selectors, namespaces) for p in pods_json])
    with NrtIngest(["n0"]) as ing:
        ing.feed_pods(json.dumps(pods_json[:30]).encode())       # appended across calls
        ing.feed_pods(json.dumps({"items": pods_json[30:]}).encode())
        pods_equal(ing.pod_objects().struct, want.struct)
        kinds = col(ing.pod_objects().struct, "ctr_kind", ing.pod_objects().struct.ctr_ptr[len(pods_json)])
        assert {0, 1} <= set(kinds)   # init and app containers (a sidecar is covered by test_pod_details)


def test_nodes(hdr):
    names = ["b", "a", "missing", "c"]
    docs = [
        {"metadata": {"name": "a", "labels": {"topology.kubernetes.io/region": "us-west-1", "topology.kubernetes.io/zone": "z2", "x": "y"}},
         "status": {"capacity": {"cpu": "64", "memory": "256Gi", "pods": "110"},
                    "allocatable": {"cpu": "63500m", "memory": "250Gi", "ephemeral-storage": "500Gi", "pods": "110", "example.com/gpu": "8", "hugepages-2Mi": "4Gi"}}},
        {"metadata": {"name": "b", "labels": {"topology.kubernetes.io/region": "us-east-1", "topology.kubernetes.io/zone": ""}},
         "status": {"capacity": {"cpu": 8}, "allocatable": {"cpu": "7", "memory": "30Gi", "requests.storage": "5"}}},
        {"metadata": {"name": "c"}, "status": {}},
        {"metadata": {"name": "somewhere-else"}, "status": {"allocatable": {"cpu": 1}}},
    ]
    with NrtIngest(names) as ing:
        assert ing.feed_nodes(json.dumps(docs).encode()) == (4, 1)
        t = ing.node_objects().struct
        assert col(t, "alloc_cpu_milli", 4) == [7000, 63500, 0, 0] and col(t, "cap_cpu_milli", 4) == [8000, 64000, 0, 0]
        assert col(t, "alloc_mem", 4) == [30 << 30, 250 << 30, 0, 0] and col(t, "alloc_eph", 4) == [0, 500 << 30, 0, 0]
        assert col(t, "alloc_pods", 4) == [0, 110, 0, 0]
        # regions / zones in first-seen order; the empty zone label of "b" counts as unset
        assert col(t, "region", 4) == [1, 0, -1, -1] and col(t, "zone", 4) == [-1, 0, -1, -1]
        assert ing.name_id("region", "us-east-1") == 1 and ing.name_id("zone", "z2") == 0
        # scalar resources: the extended and hugepage names ("requests.storage" is not a scalar name)
        gpu, hp = ing.resource_id("example.com/gpu"), ing.resource_id("hugepages-2Mi")
        assert col(t, "scalar_ptr", 5) == [0, 0, 2, 2, 2]
        assert col(t, "scalar_res", 2) == [gpu, hp] and col(t, "scalar_qty", 2) == [8, 4 << 30]
        # the same table from the Python builder
        res = O.Resources()
        res.id("example.com/gpu"), res.id("hugepages-2Mi")
        want = O.build_node_objects(hdr, res, [O.node(docs[1]["status"]["allocatable"], docs[1]["status"]["capacity"], region=1, zone=-1),
                                               O.node(docs[0]["status"]["allocatable"], docs[0]["status"]["capacity"], region=0, zone=0),
                                               O.node({}, {}), O.node({}, {})])
        for c in ("alloc_cpu_milli", "alloc_mem", "alloc_eph", "alloc_pods", "cap_cpu_milli", "region", "zone"):
            assert col(t, c, 4) == col(want.struct, c, 4), c


def test_pod_details(hdr):
    with NrtIngest(["n0"]) as ing:
        ing.seed("selector", ["a", "b", "c"])          # lexicographic ids fixed by the caller (AppGroup CR order)
        ing.seed("namespace", ["default"])
        pod = {"metadata": {"namespace": "prod", "creationTimestamp": "2024-03-01T12:00:00Z", "labels": {GROUP: "g1", SELECTOR: "c"}},
               "spec": {"priority": 1000, "overhead": {"cpu": "250m", "memory": "120Mi"},
                        "containers": [{"resources": {"limits": {"cpu": "1"}, "requests": {"cpu": "500m", "memory": "0"}}}, {"name": "no-resources"}],
                        "initContainers": [{"restartPolicy": "Always", "resources": {"requests": {"cpu": "100m"}}}, {"resources": None}]}}
        assert ing.feed_pods(json.dumps(pod).encode()) == 1
        with pytest.raises(ValueError, match="quantity"):
            ing.feed_pods(json.dumps([pod, {"spec": {"containers": [{"resources": {"requests": {"cpu": "many"}}}]}}]).encode())
        t = ing.pod_objects().struct
        assert t.n_pods == 1                                                    # the failed document left nothing behind
        assert col(t, "ctr_kind", 4) == [2, 1, 0, 0]                            # sidecar, init, app, app
        assert col(t, "req_ptr", 5) == [0, 1, 1, 3, 3] and col(t, "req_qty", 3) == [100, 500, 0]   # a zero request keeps its key
        assert col(t, "lim_ptr", 5) == [0, 0, 0, 1, 1] and col(t, "lim_qty", 1) == [1000]
        assert col(t, "ovh_qty", 2) == [250, 120 << 20] and col(t, "priority", 1) == [1000]
        assert col(t, "queue_ts", 1) == [1709294400 * 1_000_000]
        assert col(t, "selector", 1) == [2] and col(t, "ns", 1) == [1] and col(t, "appgroup", 1) == [0]
        ing.reset_pods()
        assert ing.pod_objects().struct.n_pods == 0


def test_ingested_tables_flatten_like_built_ones(hdr):
    """JSON -> decoder -> host flatteners gives the SoA columns the Python-built tables give (trimaran pods, LROC pods, Peaks pods)"""
    doc = json.loads((GOLD / "pod_manifests.json").read_text())["items"]
    res, g, s, ns = O.Resources(), O.Interner(), O.Interner(), O.Interner()
    intern_like_decoder(res, doc)
    built = O.build_pod_objects(hdr, res, [pod_json_to_dict(p, g, s, ns) for p in doc])
    L = spx.lib()
    tlp = spx.Table(hdr, "spx_tlp_params", target_utilization=40, default_requests_milli=1000, requests_multiplier=1.5)
    i64p = C.POINTER(C.c_int64)

    def flat(pods):
        a = [np.zeros(11, np.int64) for _ in range(3)]
        assert L.spx_flatten_trimaran_pods(pods.ref(), tlp.ref(), *[x.ctypes.data_as(i64p) for x in a]) == 0
        b = [np.zeros(11, np.int64) for _ in range(4)]
        assert L.spx_flatten_lroc_pods(pods.ref(), *[x.ctypes.data_as(i64p) for x in b]) == 0
        c = np.zeros(11, np.int64)
        assert L.spx_flatten_peaks_pods(pods.ref(), c.ctypes.data_as(i64p)) == 0
        return [x.tolist() for x in a + b + [c]]

    with NrtIngest(["n0"]) as ing:
        ing.feed_pods(json.dumps(doc).encode())
        assert flat(ing.pod_objects()) == flat(built)


def test_quantity_rounds_away_from_zero_and_keeps_long_mantissas():
    """resource.Quantity.Value() / MilliValue() round inexact values away from zero for either sign (negativeScaleInt64 in
    apimachinery's amount.go: value++ / value--), and apimachinery accepts more significant digits than an int64 holds"""
    from scheduler_plugins_amd.ingest import quantity
    assert quantity("-1.5", False) == -2 and quantity("-1.5", True) == -1500
    assert quantity("-100m", False) == -1 and quantity("-1", False) == -1 and quantity("-0.0001", True) == -1
    assert quantity("1.5", False) == 2 and quantity("100m", False) == 1
    assert quantity("1.00000000000000000000001", False) == 2          # 24 significant digits: sticky remainder
    assert quantity("1.00000000000000000000000", False) == 1
    assert quantity("123456789012345678901234n", False) == 123456789012346   # 1.2345...e14 rounded up
    assert quantity("12345678901234567890123", False) is None          # really out of range


def test_is_sidecar_init_container_table(hdr):
    """pkg/util/sidecar_test.go:27-61 (TestIsSidecarInitContainer) through the pod decoder: an init container is a sidecar exactly
    when its restartPolicy is "Always" — the zero value and an explicit nil are not (the NRT Filter words its rejection after the
    kind: filter.go:77-91).  The fourth row is what the API can also carry and the reference treats as not-a-sidecar: a regular
    container never is one, whatever its restartPolicy says (only initContainers are asked)."""
    cases = [({}, 1), ({"restartPolicy": None}, 1), ({"name": "init-1", "restartPolicy": "Always"}, 2)]
    with NrtIngest(["n0"]) as ing:
        pod = {"spec": {"initContainers": [c for c, _ in cases], "containers": [{"name": "app", "restartPolicy": "Always"}]}}
        assert ing.feed_pods(json.dumps(pod).encode()) == 1
        t = ing.pod_objects().struct
        assert col(t, "ctr_kind", 4) == [k for _, k in cases] + [0]
