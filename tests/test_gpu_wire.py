"""Wire formats -> host decoders -> flatteners -> GPU kernels, one test per decoder family, on the reference's own
example manifests (tests/golden/*_manifests.json; the NRT family lives in test_gpu_nrt.py::test_wire_format_to_gpu_filter):
  v1.Node + v1.Pod + AppGroup + NetworkTopology  -> k_net_cls          (NetworkOverhead Filter + Score)
  ElasticQuota                                   -> k_quota            (CapacityScheduling.PreFilter)
  load-watcher metrics                           -> k_tlp_fast2        (TargetLoadPacking Score)
Each compares the GPU tables with the CPU oracle run on the same decoded tables, plus values worked out by hand from the manifests."""
import json
from pathlib import Path

import numpy as np
import pytest

from golden import trimaran as GT
from helpers import CAPACITY, NETOVERHEAD, TLP, tlp_params
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd._abi import Table
from scheduler_plugins_amd.engine import Engine, mask_of
from scheduler_plugins_amd.ingest import NrtIngest

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
REGION, ZONE = "topology.kubernetes.io/region", "topology.kubernetes.io/zone"


def _grown(hdr, ag, group, selector, node):
    G = ag.struct.n_groups
    ptr = np.ctypeslib.as_array(ag.struct.placed_ptr, (G + 1,))
    new_ptr = [0]
    for g in range(G):
        new_ptr.append(new_ptr[-1] + int(ptr[g + 1] - ptr[g]) + sum(1 for x in group if x == g))
    assert int(ptr[G]) == 0  # decoded CRs carry no scheduled list
    order = [j for g in range(G) for j in range(len(group)) if group[j] == g]
    n = lambda f, k: np.ctypeslib.as_array(getattr(ag.struct, f), (k,)).copy() if k else np.zeros(1, np.int64)  # (an empty column may be NULL)
    nw = int(ag.struct.wl_ptr[G])
    nd = int(ag.struct.dep_ptr[nw])
    nt = int(ag.struct.topo_ptr[G])
    return Table(hdr, "spx_appgroup_objects", n_groups=G, wl_ptr=n("wl_ptr", G + 1), wl_selector=n("wl_selector", nw), dep_ptr=n("dep_ptr", nw + 1),
                 dep_selector=n("dep_selector", nd), dep_max_cost=n("dep_max_cost", nd), topo_ptr=n("topo_ptr", G + 1), topo_selector=n("topo_selector", nt),
                 topo_index=n("topo_index", nt), placed_ptr=np.array(new_ptr, np.int32), placed_selector=np.array([selector[j] for j in order], np.int32),
                 placed_node=np.array([node[j] for j in order], np.int32))


def test_network_crs_to_gpu(gpu_required, hdr, oracle):
    """manifests/appgroup/appGroup-example.yaml (a1: P1 -> P2 (cost <= 30), P2 -> P3 (cost <= 20)), redis-appGroup-example.yaml and
    networkTopology-example.yaml (regions us-west-1 / us-east-1 at 20, zones z1-z2 at 5, z3-z4 at 10) on eight nodes, two per zone;
    the scheduled list (which no CR carries: networkoverhead.go:654-694 reads the pod lister) arrives as a delta"""
    zones = [("us-west-1", "z1"), ("us-west-1", "z2"), ("us-east-1", "z3"), ("us-east-1", "z4")]
    names = [f"n{i}" for i in range(8)]
    node_docs = [{"metadata": {"name": n, "labels": {REGION: zones[i // 2][0], ZONE: zones[i // 2][1]}},
                  "status": {"allocatable": {"cpu": "8", "memory": "16Gi"}, "capacity": {"cpu": "8"}}} for i, n in enumerate(names)]
    lab = lambda g, s: {"appgroup.diktyo.x-k8s.io": g, "appgroup.diktyo.x-k8s.io.workload": s}
    pending = [("a1", "P1"), ("a1", "P2"), ("a1", "P3"), ("redis-cluster", "redis-leader"), ("redis-cluster", "redis-follower")]
    pod_docs = [{"metadata": {"namespace": "default", "name": f"p{i}", "labels": lab(g, s)}, "spec": {"containers": [{"name": "c"}]}} for i, (g, s) in enumerate(pending)]
    pod_docs.append({"metadata": {"namespace": "default", "name": "loner"}, "spec": {"containers": [{"name": "c"}]}})
    with NrtIngest(names) as ing:
        ing.feed_appgroups(json.dumps({"items": json.loads((GOLD / "appgroup_manifests.json").read_text())}).encode())
        ing.feed_nodes(json.dumps(node_docs).encode())
        ing.feed_nettopo(json.dumps(json.loads((GOLD / "nettopo_manifests.json").read_text())[0]).encode(), "UserDefined")
        ing.feed_pods(json.dumps({"items": pod_docs}).encode())
        nodes, pods, ag, nt = ing.node_objects(), ing.pod_objects(), ing.appgroup_objects(), ing.nettopo_objects()
        gid = lambda g: ing.name_id("appgroup", g)
        sid = lambda s: ing.name_id("selector", s)
        # already running: a P2 pod on n0 (z1), a P3 pod on n5 (z3), a redis follower on n2 (z2)
        placed = [("a1", "P2", 0), ("a1", "P3", 5), ("redis-cluster", "redis-follower", 2)]
        group, selector, node = [gid(g) for g, _, _ in placed], [sid(s) for _, s, _ in placed], [n for _, _, n in placed]
        with Engine(0) as e:
            e.load_c({"nodes": nodes, "pods": pods, "appgroups": ag, "nettopo": nt})  # spx_load_network: the call the cgo shim makes
            e.update_net_placed(e.flatten_net_placed(pods, ag, group, selector, node))
            e.eval(mask_of(NETOVERHEAD))
            e.sync()
            status, scores = e.all_status(NETOVERHEAD), e.all_scores(NETOVERHEAD)
            cost = np.stack([e.raw(NETOVERHEAD, p, 0) for p in range(len(pod_docs))])
        want = oracle.Snapshot(nodes, pods, appgroups=_grown(hdr, ag, group, selector, node), nettopo=nt)
        assert np.array_equal(status, want.filter_rows(NETOVERHEAD))
        raw, norm = want.score_rows(NETOVERHEAD)
        assert np.array_equal(cost, raw)
        assert np.array_equal(scores.astype(np.int64)[status == 0], norm[status == 0])
    # by hand.  p0 (P1) depends on P2 running on n0: same host 0, same zone (n1) 1, z1->z2 5, other region 20 — all within 30
    assert cost[0].tolist() == [0, 1, 5, 5, 20, 20, 20, 20] and not status[0].any()
    # p1 (P2) depends on P3 on n5 (z3): same host 0, same zone (n4) 1, z4 10, the other region 20 — all <= 20: nothing violated
    assert cost[1].tolist() == [20, 20, 20, 20, 1, 0, 10, 10] and not status[1].any()
    # p2 (P3) has no dependencies, the loner no AppGroup: they score equally everywhere
    assert not cost[2].any() and not cost[5].any() and len(set(scores[2].tolist())) == 1 and len(set(scores[5].tolist())) == 1
    # p3 (redis-leader) depends on the follower on n2 (z2) with cost <= 80
    assert cost[3].tolist() == [5, 5, 0, 1, 20, 20, 20, 20]


def test_elasticquota_cr_to_gpu(gpu_required, hdr, oracle):
    """manifests/capacityscheduling/elasticquota-example.yaml (namespace test: min cpu 10 / 20Gi / 1 gpu, max cpu 20 / 40Gi / 2 gpus)
    served with a status.used, a second quota beside it; CapacityScheduling.PreFilter per pending pod on the GPU"""
    example = {"apiVersion": "scheduling.x-k8s.io/v1alpha1", "kind": "ElasticQuota", "metadata": {"name": "test", "namespace": "test"},
               "spec": {"max": {"cpu": 20, "memory": "40Gi", "nvidia.com/gpu": 2}, "min": {"cpu": 10, "memory": "20Gi", "nvidia.com/gpu": 1}},
               "status": {"used": {"cpu": 18, "memory": "10Gi", "nvidia.com/gpu": 1}}}
    other = {"apiVersion": "scheduling.x-k8s.io/v1alpha1", "kind": "ElasticQuota", "metadata": {"name": "dev", "namespace": "dev"},
             "spec": {"max": {"cpu": 16, "memory": "16Gi"}, "min": {"cpu": 12, "memory": "8Gi"}}, "status": {"used": {"cpu": 1, "memory": "1Gi"}}}
    namespaces = ["default", "test", "dev"]
    res = O.Resources()
    res.id("nvidia.com/gpu")
    pods = O.build_pod_objects(hdr, res, [
        O.pod([O.container({"cpu": "3", "memory": "1Gi"})], ns=1),                       # 18 + 3 > max 20                     -> over max
        O.pod([O.container({"cpu": "1", "nvidia.com/gpu": "2"})], ns=1),                 # 1 + 2 gpus > max 2                  -> over max
        O.pod([O.container({"cpu": "1", "memory": "1Gi"})], ns=1),                       # within max, aggregate 19 + 1 <= sum(min) 22 -> passes
        O.pod([O.container({"cpu": "4", "memory": "1Gi"})], ns=2),                       # within dev's max, aggregate 19 + 4 > 22     -> over min
        O.pod([O.container({"cpu": "64"})], ns=0),                                       # no quota in the namespace            -> passes
    ])
    with NrtIngest(["n0"]) as ing:
        assert ing.feed_quotas(json.dumps({"items": [example, other]}).encode(), namespaces)[0] == 2
        quota = ing.quota_objects()
        with Engine(0) as e:
            e.load_c({"pods": pods, "rc": res.table(hdr), "quota": quota})  # spx_load_quota
            e.eval(mask_of(CAPACITY))
            e.sync()
            got = e.prefilter(CAPACITY).tolist()
        f = oracle.lib().orc_capacity_prefilter
        assert got == [f(pods.ref(), res.table(hdr).ref(), quota.ref(), i) for i in range(5)]
    OVER_MAX, OVER_MIN = hdr.consts["SPX_QUOTA_ST_OVER_MAX"], hdr.consts["SPX_QUOTA_ST_OVER_MIN"]
    assert got == [OVER_MAX, OVER_MAX, 0, OVER_MIN, 0]


@pytest.mark.parametrize("case", GT.TLP_CASES, ids=lambda c: f"L{c['line']}")
def test_watcher_metrics_to_gpu(gpu_required, hdr, case):
    """targetloadpacking_test.go:148-238 with the load-watcher response decoded from its wire format (json.Marshal of
    watcher.WatcherMetrics, rendered by tests/test_ingest_metrics.py::marshal) and scored by k_tlp_fast2"""
    from test_ingest_metrics import marshal
    res = O.Resources()
    nodes = O.build_node_objects(hdr, res, [O.node(GT.NODE)])
    pods = O.build_pod_objects(hdr, res, [case["pod"]])
    with NrtIngest(["node-1"]) as ing:
        ing.feed_metrics(marshal(None if case["metrics"] is None else {"node-1": ms for _, ms in case["metrics"].items()}))
        with Engine(0) as e:
            e.set_tlp(**GT.TLP_PARAMS)
            e.load_trimaran_objects(nodes, res.table(hdr), pods, ing.metrics_objects())
            e.eval(mask_of(TLP))
            e.sync()
            assert e.scores(TLP, 0).tolist() == case["expected"] and e.raw(TLP, 0).tolist() == case["expected"]
