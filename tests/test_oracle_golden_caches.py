"""The reference's own tests of the two control-plane caches whose verdicts feed the hot path, replayed from the transcribed data
(tests/golden/nrt_discard_reserved.json <- cache/discardreserved_test.go:34-140; tests/golden/trimaran_handler.json <-
pkg/trimaran/handler_test.go:12-77) through oracle/cache_models.py, and tied to what the product consumes: the `fresh` column
(NRT Filter: "invalid node topology data") and the assigned-pod list TargetLoadPacking's flattener walks."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

from helpers import NRT, tlp_params
from scheduler_plugins_amd import objects as O

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT.parent / "oracle"))
from cache_models import DiscardReservedModel, PodAssignHandlerModel  # noqa: E402

DISCARD = json.loads((ROOT / "golden" / "nrt_discard_reserved.json").read_text())
HANDLER = json.loads((ROOT / "golden" / "trimaran_handler.json").read_text())


@pytest.mark.parametrize("test", DISCARD["tests"], ids=lambda t: t["name"])
def test_discard_reserved_reference_tests(test):
    m = DiscardReservedModel()
    for step in test["steps"]:
        op = step["op"]
        if op == "store_nrt":
            m.api[step["node"]] = {"name": step["node"]}
        elif op == "preset":
            m.reservation_map = {step["node"]: {step["uid"]: True}}
        elif op == "reserve":
            m.reserve(step["node"], step["uid"])
        elif op == "remove":
            m.remove_reservation(step["node"], step["uid"])
        elif op == "get":
            nrt, fresh = m.get_cached_nrt_copy(step["node"])
            assert fresh == step["expect_ok"] and (nrt is not None) == step["expect_nrt"]
        elif op == "expect_map":
            assert step["node"] in m.reservation_map and m.reservation_map[step["node"]] == step["uids"]
        else:
            raise AssertionError(op)


def test_discard_reserved_verdict_reaches_the_filter(hdr, oracle):
    """the scenario of TestDiscardReservedNodesGetNRTCopyFails + ...RemoveReservationForNode on a two-node snapshot: the reserved node
    answers Fresh == false -> Filter "invalid node topology data" for every filtered pod (filter.go:197-199); after the removal the
    node is evaluated from its NRT again"""
    res = O.Resources()
    zones = [{"name": "node-0", "resources": {"cpu": "4", "memory": "8Gi"}}, {"name": "node-1", "resources": {"cpu": "4", "memory": "8Gi"}}]
    names = ["node1", "node2"]
    nodes = O.build_node_objects(hdr, res, [O.node_from_zones(zones)] * 2)
    pods = O.build_pod_objects(hdr, res, [O.pod([O.container({"cpu": "2", "memory": "1Gi"}, {"cpu": "2", "memory": "1Gi"})])])
    params = O.nrt_params(hdr, res, "LeastAllocated")
    m = DiscardReservedModel({n: O.nrt(zones, policies=["SingleNUMANodeContainerLevel"]) for n in names})
    INVALID = hdr.consts["SPX_NRT_ST_INVALID_TOPOLOGY"]

    def statuses():
        view = [m.get_cached_nrt_copy(n) for n in names]
        nrt_t = O.build_nrt_objects(hdr, res, [v[0] for v in view], fresh=[v[1] for v in view])
        snap = oracle.Snapshot(nodes, pods, rc=res.table(hdr), nrt=nrt_t, nrt_params=params)
        return snap.filter_rows(NRT)[0].tolist()

    assert statuses() == [0, 0]
    step = DISCARD["tests"][3]["steps"]
    m.reserve(step[0]["node"], step[0]["uid"])
    assert statuses() == [INVALID, 0]
    m.remove_reservation(step[2]["node"], step[2]["uid"])
    assert statuses() == [0, 0]


@pytest.mark.parametrize("case", HANDLER["cases"], ids=lambda c: c["name"])
def test_handler_cache_cleanup_reference_cases(case):
    now = 1_700_000_000.0
    node = HANDLER["node"]
    h = PodAssignHandlerModel(HANDLER["reporting_interval_s"])
    h.cache[node] = [(None if e["age_offset_s"] is None else now + e["age_offset_s"], e["pod"]) for e in case["cache"]]
    if case["pod_to_update"]:
        h.on_update("", node, case["pod_to_update"], now)  # oldPod has no NodeName, newPod is assigned to testNode (:63-68)
    h.cleanup(now)
    assert h.pods(node) == case["expected_pods"] and len(h.pods(node)) == case["expected_size"]


def test_cleanup_and_score_agree_on_which_pods_are_recent(hdr):
    """handler_test.go's third case through the product's flattener: with Window.End == now, the pods cleanupCache keeps are the pods
    TargetLoadPacking's Score adds to the node's load (targetloadpacking.go:153-160: bound after the window, or less than
    metricsAgentReportingIntervalSeconds before its end) — the same 60 s on both sides."""
    import ctypes as C
    import scheduler_plugins_amd as spx
    case = HANDLER["cases"][2]
    now = 1_700_000_000
    h = PodAssignHandlerModel(HANDLER["reporting_interval_s"])
    h.cache["n"] = [(float(now + e["age_offset_s"]), e["pod"]) for e in case["cache"]]
    h.cleanup(float(now))
    res = O.Resources()
    milli = {"Pod-1": 100, "Pod-2": 200, "Pod-3": 400}
    nodes = O.build_node_objects(hdr, res, [O.node({"cpu": "8", "memory": "32Gi"})])
    metrics = O.build_metrics_objects(hdr, 1, {0: [("CPU", "AVG", 10)]}, window_end=now)
    assigned = O.build_assigned_objects(hdr, res, 1, {0: [(now + e["age_offset_s"], O.pod([O.container(limits={"cpu": f"{milli[e['pod']]}m"})])) for e in case["cache"]]})
    tlp = tlp_params(hdr, 40, 1000, 1.5)
    missing = np.zeros(1, np.int64)
    fn = spx.lib().spx_flatten_trimaran_nodes
    args = [None] * 11
    args[2] = missing.ctypes.data_as(fn.argtypes[6])
    assert fn(nodes.ref(), metrics.ref(), assigned.ref(), tlp.ref(), *args) == 0
    assert missing[0] == sum(milli[p] for p in h.pods("n")) == 600
