"""Wire-format ingestion of ElasticQuota CRs: JSON -> spx_quota_objects against the independent Python
builder, on the reference's example (manifests/capacityscheduling/elasticquota-example.yaml, inlined below: 16 lines) and on
quotas shaped like its unit-test fixtures; then CapacityScheduling.PreFilter through the oracle on the decoded table.  CPU only."""
import json

import numpy as np
import pytest

from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd.ingest import NrtIngest

# manifests/capacityscheduling/elasticquota-example.yaml as the API server would serve it
EXAMPLE = {"apiVersion": "scheduling.x-k8s.io/v1alpha1", "kind": "ElasticQuota", "metadata": {"name": "test", "namespace": "test"},
           "spec": {"max": {"cpu": 20, "memory": "40Gi", "nvidia.com/gpu": 2}, "min": {"cpu": 10, "memory": "20Gi", "nvidia.com/gpu": 1}}}


def col(struct, name, n):
    return np.ctypeslib.as_array(getattr(struct, name), (n,)).tolist() if n else []


def quotas_equal(a, b, n):
    assert (a.n_namespaces, a.n_scalar_slots) == (b.n_namespaces, b.n_scalar_slots)
    assert col(a, "scalar_res", 4)[:a.n_scalar_slots] == col(b, "scalar_res", 4)[:b.n_scalar_slots]
    for c in ("has_quota", "min_present", "max_present", "used_present"):
        assert col(a, c, n) == col(b, c, n), c
    for c in ("min", "max", "used"):
        assert col(a, c, n * 8) == col(b, c, n * 8), c


def cr(ns, q):
    out = {"apiVersion": "scheduling.x-k8s.io/v1alpha1", "kind": "ElasticQuota", "metadata": {"name": f"q-{ns}", "namespace": ns}, "spec": {}}
    if q.get("min") is not None:
        out["spec"]["min"] = q["min"]
    if q.get("max") is not None:
        out["spec"]["max"] = q["max"]
    if q.get("used") is not None:
        out["status"] = {"used": q["used"]}
    return out


def test_reference_example(hdr):
    namespaces = ["default", "test"]
    res = O.Resources()
    res.id("nvidia.com/gpu")
    want = O.build_quota_objects(hdr, res, [None, {"min": EXAMPLE["spec"]["min"], "max": EXAMPLE["spec"]["max"], "used": None}])
    with NrtIngest(["n0"]) as ing:
        assert ing.feed_quotas(json.dumps(EXAMPLE).encode(), namespaces) == (1, 0)
        t = ing.quota_objects().struct
        quotas_equal(t, want.struct, 2)
        assert col(t, "min", 16)[8:13] == [10_000, 20 << 30, 0, 0, 1] and col(t, "max", 16)[8:13] == [20_000, 40 << 30, 0, 0, 2]
        assert col(t, "max", 16)[:3] == [(1 << 63) - 1] * 3           # a namespace without a quota: the nil-list bounds
        assert ing.name_id("namespace", "test") == 1


def test_varied_quotas_match_builder(hdr):
    quotas = {
        "a": {"min": {"cpu": "500m", "memory": "1Gi"}, "max": {"cpu": "4", "memory": "8Gi", "pods": "10"}, "used": {"cpu": "250m", "memory": "100Mi"}},
        "b": {"min": None, "max": None, "used": {"example.com/gpu": "3", "cpu": "1"}},                       # nil min / max
        "c": {"min": {"example.com/gpu": "1", "hugepages-2Mi": "1Gi"}, "max": {"example.com/gpu": "8"}, "used": None},
        "d": {"min": {"cpu": "1", "ephemeral-storage": "10Gi", "requests.storage": "5"}, "max": {"cpu": "2"}, "used": {}},  # requests.* is not a scalar name
    }
    namespaces = ["c", "none", "a", "d", "b"]
    docs = [cr(ns, quotas[ns]) for ns in ("a", "b", "c", "d")] + [cr("elsewhere", quotas["a"])]
    res = O.Resources()
    for d in docs:  # resource ids in the order the decoder meets the names
        for part in (d["spec"].get("min"), d["spec"].get("max"), (d.get("status") or {}).get("used")):
            for r in part or {}:
                res.id(r)
    # the builder assigns scalar slots in namespace order, the decoder too (rows are laid out by namespace index)
    want = O.build_quota_objects(hdr, res, [quotas.get(ns) for ns in namespaces])
    with NrtIngest(["n0"]) as ing:
        assert ing.feed_quotas(json.dumps({"items": docs}).encode(), namespaces) == (5, 1)
        quotas_equal(ing.quota_objects().struct, want.struct, len(namespaces))


def test_prefilter_on_decoded_quota(hdr, oracle):
    """capacity_scheduling.go:208-283 through the oracle: a pod over its namespace's max is rejected, one within it passes"""
    namespaces = ["team-a", "team-b"]
    docs = [cr("team-a", {"min": {"cpu": "2", "memory": "4Gi"}, "max": {"cpu": "4", "memory": "8Gi"}, "used": {"cpu": "3", "memory": "1Gi"}}),
            cr("team-b", {"min": {"cpu": "6", "memory": "4Gi"}, "max": {"cpu": "8", "memory": "16Gi"}, "used": {"cpu": "4", "memory": "1Gi"}})]
    res = O.Resources()
    pods = O.build_pod_objects(hdr, res, [O.pod([O.container({"cpu": "2", "memory": "1Gi"})], ns=0),      # 3 + 2 > max 4 of team-a
                                          O.pod([O.container({"cpu": "500m", "memory": "1Gi"})], ns=0),   # total used 7 + 0.5 <= total min 8
                                          O.pod([O.container({"cpu": "1500m", "memory": "1Gi"})], ns=1)]) # within team-b's max, but 7 + 1.5 > total min 8
    with NrtIngest(["n0"]) as ing:
        ing.feed_quotas(json.dumps(docs).encode(), namespaces)
        f = oracle.lib().orc_capacity_prefilter
        got = [f(pods.ref(), res.table(hdr).ref(), ing.quota_objects().ref(), i) for i in range(3)]
        want_t = O.build_quota_objects(hdr, res, [{"min": d["spec"]["min"], "max": d["spec"]["max"], "used": d["status"]["used"]} for d in docs])
        assert got == [f(pods.ref(), res.table(hdr).ref(), want_t.ref(), i) for i in range(3)]
        assert got == [1, 0, 2]   # SPX_QUOTA_ST_OVER_MAX, pass, SPX_QUOTA_ST_OVER_MIN (the aggregate check, capacity_scheduling.go:279)


def test_nominated_pods_come_from_the_pod_table(hdr, oracle):
    """capacity_scheduling.go:231-253: PreFilter adds the requests of nominated pods (PodNominator.NominatedPodsForNode over the
    snapshot's nodes).  On the wire a nominated pod is a pending pod with status.nominatedNodeName; the decoder derives the list
    from the pod table, and the oracle's PreFilter answers exactly as with the Python builder's list."""
    namespaces = ["team-a", "team-b"]
    docs = [cr("team-a", {"min": {"cpu": "2", "memory": "4Gi"}, "max": {"cpu": "4", "memory": "8Gi"}, "used": {"cpu": "1", "memory": "1Gi"}}),
            cr("team-b", {"min": {"cpu": "6", "memory": "4Gi"}, "max": {"cpu": "8", "memory": "16Gi"}, "used": {"cpu": "2", "memory": "1Gi"}})]

    def pod_doc(ns, cpu, prio, nominated=None):
        d = {"metadata": {"namespace": ns}, "spec": {"priority": prio, "containers": [{"name": "c", "resources": {"requests": {"cpu": cpu, "memory": "1Gi"}}}]}}
        if nominated is not None:
            d["status"] = {"nominatedNodeName": nominated}
        return d
    pend = [pod_doc("team-a", "1500m", 10),                        # 1 + 1.5 + nominated 2 (row 1, same quota, higher priority) > max 4
            pod_doc("team-a", "2", 100, nominated="n1"),           # nominated itself: its own request is not added twice
            pod_doc("team-b", "1", 5, nominated="gone"),           # nominated to a node outside the snapshot: not in the list
            pod_doc("team-a", "1500m", 1000)]                      # more important than the nominated pod: does not count it
    res = O.Resources()
    mk = lambda ns, cpu, prio: O.pod([O.container({"cpu": cpu, "memory": "1Gi"})], ns=ns, priority=prio)
    pods_t = O.build_pod_objects(hdr, res, [mk(0, "1500m", 10), mk(0, "2", 100), mk(1, "1", 5), mk(0, "1500m", 1000)])
    want_q = O.build_quota_objects(hdr, res, [{"min": d["spec"]["min"], "max": d["spec"]["max"], "used": d["status"]["used"]} for d in docs],
                                   nominated=[(0, 100, 1, mk(0, "2", 100))])
    with NrtIngest(["n0", "n1"]) as ing:
        ing.feed_quotas(json.dumps(docs).encode(), namespaces)
        assert ing.quota_objects().struct.n_nominated == 0
        ing.feed_pods(json.dumps(pend).encode())
        q = ing.quota_objects().struct
        assert q.n_nominated == 1 and q.nom_ns[0] == 0 and q.nom_priority[0] == 100 and q.nom_pending_index[0] == 1
        f = oracle.lib().orc_capacity_prefilter
        rc = ing.resource_classes()
        got = [f(ing.pod_objects().ref(), rc.ref(), ing.quota_objects().ref(), i) for i in range(4)]
        want = [f(pods_t.ref(), res.table(hdr).ref(), want_q.ref(), i) for i in range(4)]
        assert got == want == [1, 0, 0, 0]
        ing.reset_pods()
        assert ing.quota_objects().struct.n_nominated == 0
