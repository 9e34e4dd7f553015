"""load-watcher response -> spx_metrics_objects (trimaran Collector.updateMetrics, collector.go:139-150).

PARITY UNPINNED: the struct the reference decodes into (watcher.WatcherMetrics, github.com/paypal/load-watcher v0.2.4) is not
vendored and the reference holds no golden document — its tests json.Marshal the Go struct inside httptest servers.  These tests
therefore render the reference's own test metrics (tests/golden/trimaran.py) the way json.Marshal renders that struct (published
json tags; Data.NodeMetricsMap has no tag, hence its Go name), decode them, and require (a) the decoded columns to equal the
independent Python builder's and (b) the oracle to reproduce the reference's expected scores from the decoded table.  The draft
payload of KEP-61 (kep/61-Trimaran-real-load-aware-scheduling/README.md:301-352, committed as a fixture) has a different shape;
decoded with encoding/json's rules into the real struct it leaves NodeMetricsMap nil, which is what the decoder must report."""
import json
from pathlib import Path

import numpy as np
import pytest

from golden import trimaran as GT
from helpers import TLP, tlp_params
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd.ingest import NrtIngest

KEP61 = Path(__file__).parent / "golden" / "loadwatcher_kep61_sample.json"


def marshal(node_metrics, window_end=1_700_000_000):
    """json.Marshal(watcher.WatcherMetrics{...}) for {node name: [(type, operator, value), ...] | None}"""
    nm = None
    if node_metrics is not None:
        nm = {n: {"metrics": None if ms is None else [{"name": "", "type": t, "operator": o, "value": v} for t, o, v in ms], "tags": {}, "metadata": {}}
              for n, ms in node_metrics.items()}
    return json.dumps({"timestamp": window_end + 60, "window": {"duration": "15m", "start": window_end - 900, "end": window_end}, "source": "test",
                       "data": {"NodeMetricsMap": nm}}).encode()


def cols(t, n):
    g = lambda name, k: np.ctypeslib.as_array(getattr(t, name), (k,)).tolist() if k else []
    m = t.m_ptr[n]
    return dict(nil=t.map_is_nil, end=t.window_end, present=g("node_present", n), isnil=g("node_metrics_nil", n), ptr=g("m_ptr", n + 1),
                type=g("m_type", m), op=g("m_op", m), value=g("m_value", m))


def test_decodes_like_the_builder(hdr):
    names = ["n0", "n1", "n2", "n3", "n4"]
    data = {"n0": [("CPU", "AVG", 20.5), ("CPU", "STD", 3.25), ("Memory", "Latest", 40)], "n2": [], "n3": None,
            "n4": [("CPU", "", 7), ("cpu", "AVG", 1), ("Memory", "SUM", 2.5e-3)], "stranger": [("CPU", "AVG", 1)]}
    want = O.build_metrics_objects(hdr, 5, {0: data["n0"], 2: [], 3: None, 4: data["n4"]}, window_end=1_700_000_000)
    with NrtIngest(names) as ing:
        assert ing.metrics_objects() is None
        assert ing.feed_metrics(marshal(data)) == (5, 1)
        got = cols(ing.metrics_objects().struct, 5)
        assert got == cols(want.struct, 5)
        assert got["type"][-2:] == [hdr.consts["SPX_MT_OTHER"]] * 1 + [hdr.consts["SPX_MT_MEMORY"]]   # "cpu" != watcher.CPU
        assert got["op"][-3:] == [hdr.consts["SPX_MO_EMPTY"], hdr.consts["SPX_MO_AVG"], hdr.consts["SPX_MO_OTHER"]]
        # encoding/json matches member names case-insensitively and ignores unknown members
        doc = json.loads(marshal({"n1": [("CPU", "Latest", 9)]}))
        doc["data"] = {"nodemetricsmap": doc["data"]["NodeMetricsMap"], "extra": [1, 2, {"x": None}]}
        doc["WINDOW"] = doc.pop("window")
        ing.feed_metrics(json.dumps(doc).encode())
        got = cols(ing.metrics_objects().struct, 5)
        assert got["present"] == [0, 1, 0, 0, 0] and got["value"] == [9.0] and got["end"] == 1_700_000_000   # a fetch replaces the snapshot
        # a failed fetch keeps the previous metrics (collector.go:140-150)
        with pytest.raises(ValueError):
            ing.feed_metrics(b'{"data": {"NodeMetricsMap": {"n0": {"metrics": [{"value": "x"')
        assert cols(ing.metrics_objects().struct, 5)["value"] == [9.0]
        ing.feed_metrics(marshal(None))
        assert ing.metrics_objects().struct.map_is_nil == 1


def test_kep61_draft_payload_leaves_the_map_nil(hdr):
    with NrtIngest(["node-1", "node-2"]) as ing:
        assert ing.feed_metrics(KEP61.read_bytes()) == (0, 0)
        t = ing.metrics_objects().struct
        assert t.map_is_nil == 1 and t.window_end == 1556985422


@pytest.mark.parametrize("case", GT.TLP_CASES, ids=lambda c: f"L{c['line']}")
def test_decoded_metrics_reproduce_the_reference_tlp_scores(hdr, oracle, case):
    """targetloadpacking_test.go:148-238 with the watcher response going through the wire format"""
    res = O.Resources()
    nodes = O.build_node_objects(hdr, res, [O.node(GT.NODE)])
    pods = O.build_pod_objects(hdr, res, [case["pod"]])
    with NrtIngest(["node-1"]) as ing:
        ing.feed_metrics(marshal(None if case["metrics"] is None else {"node-1": ms for _, ms in case["metrics"].items()}))
        snap = oracle.Snapshot(nodes, pods, rc=res.table(hdr), metrics=ing.metrics_objects(), tlp_params=tlp_params(hdr, **GT.TLP_PARAMS))
        raw, _ = snap.score_rows(TLP)
        assert raw[0].tolist() == case["expected"]


def test_collector_fixtures_round_trip(hdr):
    """pkg/trimaran/collector_test.go: TestGetNodeMetrics (:149-170, fixture watcherResponse :38-68) — what the server marshalled is
    what GetNodeMetrics returns for the node — and TestGetNodeMetricsNilForNode (:172-194, noWatcherResponseForNode :70-75): an
    empty NodeMetricsMap is a non-nil response with no metrics for the node."""
    response = {"node-1": [("CPU", "AVG", 80), ("CPU", "STD", 16), ("Memory", "AVG", 25), ("Memory", "STD", 6.25)]}
    K = hdr.consts
    with NrtIngest(["node-1", "node-2"]) as ing:
        assert ing.feed_metrics(marshal(response)) == (1, 0)   # one map entry, none outside the snapshot
        got = cols(ing.metrics_objects().struct, 2)
        assert got["nil"] == 0 and got["present"] == [1, 0] and got["ptr"] == [0, 4, 4]
        assert got["type"] == [K["SPX_MT_CPU"], K["SPX_MT_CPU"], K["SPX_MT_MEMORY"], K["SPX_MT_MEMORY"]]
        assert got["op"] == [K["SPX_MO_AVG"], K["SPX_MO_STD"], K["SPX_MO_AVG"], K["SPX_MO_STD"]]
        assert got["value"] == [80.0, 16.0, 25.0, 6.25]
        assert ing.feed_metrics(marshal({})) == (0, 0)
        got = cols(ing.metrics_objects().struct, 2)
        assert got["nil"] == 0 and got["present"] == [0, 0] and got["value"] == []
