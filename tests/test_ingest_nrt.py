"""Wire-format ingestion (first slice): NodeResourceTopology JSON -> spx_nrt_objects, in the product's
host library (CPU only).  Pins: the reference's example manifests (tests/golden/nrt_manifests.json), the NRTs of its
integration table (nrt_integration.json) and the quantity semantics of SURVEY appendix A, each against the independent
Python builders of scheduler_plugins_amd.objects; then the ingested tables drive the oracle to the same Filter/Score results."""
import json
import time
from fractions import Fraction
from pathlib import Path

import numpy as np
import pytest

from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd.ingest import NrtIngest, quantity

GOLD = Path(__file__).parent / "golden"
COLS = ("has_nrt", "fresh", "legacy_policy", "attr_scope", "attr_policy", "attr_max_numa", "zone_ptr", "zone_is_node", "zone_numa_id",
        "zres_ptr", "zres_res", "zres_avail", "zcost_ptr", "zcost_numa_id", "zcost_value", "zres_allocatable")


def column(struct, name, n):
    return np.ctypeslib.as_array(getattr(struct, name), (n,)).tolist() if n else []


def tables_equal(a, b):
    """a, b: spx_nrt_objects structs"""
    n = a.n_nodes
    assert n == b.n_nodes
    nz = a.zone_ptr[n]
    ne, nc = a.zres_ptr[nz], a.zcost_ptr[nz]
    sizes = dict(has_nrt=n, fresh=n, legacy_policy=n, attr_scope=n, attr_policy=n, attr_max_numa=n, zone_ptr=n + 1, zone_is_node=nz,
                 zone_numa_id=nz, zres_ptr=nz + 1, zres_res=ne, zres_avail=ne, zcost_ptr=nz + 1, zcost_numa_id=nc, zcost_value=nc,
                 zres_allocatable=ne)
    for c in COLS:
        assert column(a, c, sizes[c]) == column(b, c, sizes[c]), c


def cr_to_dict(cr):
    """the same CR as the dict form objects.nrt() takes (independent of the C++ decoder)"""
    zones = []
    for z in cr.get("zones", []):
        zones.append({"name": z["name"], "type": z.get("type", ""),
                      "resources": [(r["name"], r.get("capacity", r["available"]), r.get("allocatable", r["available"]), r["available"])
                                    for r in z.get("resources", [])],
                      "costs": [(c["name"], c["value"]) for c in z.get("costs", [])]})
    return O.nrt(zones, cr.get("topologyPolicies", []), {a["name"]: a["value"] for a in cr.get("attributes", [])})


def test_reference_manifests(hdr):
    docs = json.loads((GOLD / "nrt_manifests.json").read_text())
    names = ["worker-node-b", "some-node-without-nrt", "worker-node-a"]
    with NrtIngest(names) as ing:
        n, unknown = ing.feed(json.dumps({"kind": "NodeResourceTopologyList", "items": docs}).encode())
        assert (n, unknown) == (2, 0)
        res = O.Resources()
        by_name = {d["metadata"]["name"]: d for d in docs}
        # intern in the order the decoder meets the names: worker-node-a's resources first (document order)
        for d in docs:
            for z in d["zones"]:
                for r in z["resources"]:
                    res.id(r["name"])
        want = O.build_nrt_objects(hdr, res, [cr_to_dict(by_name[x]) if x in by_name else None for x in names])
        tables_equal(ing.nrt_objects().struct, want.struct)
        assert ing.resource_id("example.com/deviceA") == res.ids["example.com/deviceA"]
        fl = ing.resource_classes().struct
        assert column(fl, "flags", fl.n_res) == res.flags().tolist()
        # worker-node-A.yaml: zone node-0 has cpu 3 available (millicores), deviceA 1, deviceB 2
        t = ing.nrt_objects().struct
        z0 = t.zone_ptr[2]
        assert column(t, "zres_avail", t.zres_ptr[t.zone_ptr[3]])[t.zres_ptr[z0]:t.zres_ptr[z0 + 1]] == [3000, 1, 2]


def test_integration_table_nrts_round_trip(hdr):
    """the 29 NRT pairs of test/integration/noderesourcetopology_test.go, rendered as CR JSON and decoded again"""
    cases = json.loads((GOLD / "nrt_integration.json").read_text())["cases"]
    checked = 0
    for case in cases:
        if not case["nrts"]:
            continue
        crs, dicts, names = [], [], []
        res = O.Resources()
        for i, n in enumerate(case["nrts"]):
            name = n.get("name", f"node-{i}")
            names.append(name)
            zones = [{"name": z["name"], "type": z.get("type", "Node"),
                      "resources": [{"name": k, "capacity": str(c), "allocatable": str(a), "available": str(a)} for k, c, a in z["resources"]],
                      "costs": [{"name": k, "value": v} for k, v in (z["costs"].items() if isinstance(z.get("costs"), dict) else (z.get("costs") or []))]}
                     for z in n["zones"]]
            cr = {"apiVersion": "topology.node.k8s.io/v1alpha2", "kind": "NodeResourceTopology", "metadata": {"name": name}, "zones": zones,
                  "topologyPolicies": n.get("policies", []),
                  "attributes": [{"name": k, "value": str(v)} for k, v in (n.get("attributes") or {}).items()]}
            crs.append(cr)
            dicts.append(cr_to_dict(cr))
        with NrtIngest(names) as ing:
            ing.feed(json.dumps(crs).encode())
            for cr in crs:
                for z in cr["zones"]:
                    for r in z["resources"]:
                        res.id(r["name"])
            tables_equal(ing.nrt_objects().struct, O.build_nrt_objects(hdr, res, dicts).struct)
            checked += 1
    assert checked >= 25


QUANTITIES = ["0", "1", "3", "100m", "1500m", "0.5", "1.5", "2.0001", "250u", "1n", "999999n", "4Gi", "500Mi", "1Ki", "2Ti", "1.5Gi", "0.1Ki",
              "1e3", "1E3", "1e-3", "12e2", "1.25e1", "5k", "7M", "1G", "2T", "1P", "1E", "1Ei", "3Pi", "100", "+7", "-5", "-1500m", "-0.5",
              "123456789", "9223372036854775807", "0.000", "10.", ".5"]


@pytest.mark.parametrize("text", QUANTITIES)
def test_quantity_semantics(text):
    """MilliValue() / Value() round inexact values away from zero (a ceiling for v >= 0), against exact rationals"""
    try:
        fr = O.parse_quantity(text)
    except (ValueError, ZeroDivisionError):
        fr = None
    for milli in (False, True):
        got = quantity(text, milli)
        if fr is None:
            assert got is None
            continue
        want = O._ceil(fr * 1000) if milli else O._ceil(fr)
        if abs(want) > (1 << 63) - 1:
            assert got is None
        else:
            assert got == want, (text, milli)


@pytest.mark.parametrize("text", ["", "abc", "1x", "1ee3", "Gi", "1.2.3", "--1", "1e", "5 Gi"])
def test_quantity_rejects_garbage(text):
    assert quantity(text, False) is None


def test_errors_and_updates(hdr):
    with NrtIngest(["a", "b"]) as ing:
        with pytest.raises(ValueError, match="JSON"):
            ing.feed(b'{"items": [')
        with pytest.raises(ValueError, match="metadata.name"):
            ing.feed(b'[{"zones": []}]')
        with pytest.raises(ValueError, match="available"):
            ing.feed(b'{"metadata": {"name": "a"}, "zones": [{"name": "node-0", "type": "Node", "resources": [{"name": "cpu", "available": "lots"}]}]}')
        one = {"metadata": {"name": "a"}, "zones": [{"name": "node-0", "type": "Node", "resources": [{"name": "cpu", "available": "2", "allocatable": 4, "capacity": 4}]}]}
        assert ing.feed(json.dumps(one).encode()) == (1, 0)
        assert ing.feed(json.dumps({"metadata": {"name": "zzz"}}).encode()) == (1, 1)      # not a node of the snapshot
        t = ing.nrt_objects().struct
        assert column(t, "has_nrt", 2) == [1, 0] and column(t, "zres_avail", 1) == [2000] and column(t, "zres_allocatable", 1) == [4000]
        one["zones"][0]["resources"][0]["available"] = "1500m"                              # a watch event replaces the object
        one["zones"][0]["name"] = "node-\\u0031".encode().decode("unicode_escape")          # plain "node-1"
        ing.feed(json.dumps(one).encode())
        t = ing.nrt_objects().struct
        assert column(t, "zres_avail", 1) == [1500] and column(t, "zone_numa_id", 1) == [1]


def test_ingested_tables_drive_the_same_filter_and_score(hdr, oracle):
    """end to end on the CPU: CR JSON -> C++ decoder -> oracle Filter/Score == Python builder -> oracle Filter/Score"""
    docs = json.loads((GOLD / "nrt_manifests.json").read_text())
    names = [d["metadata"]["name"] for d in docs]
    res = O.Resources()
    for d in docs:
        for z in d["zones"]:
            for r in z["resources"]:
                res.id(r["name"])
    want_t = O.build_nrt_objects(hdr, res, [cr_to_dict(d) for d in docs])
    nodes = O.build_node_objects(hdr, res, [O.node({"cpu": "8", "memory": "16Gi", "example.com/deviceA": "3", "example.com/deviceB": "3"})] * 2)
    # Burstable pods (no limits): only the devices are NUMA-affine for them.  The manifests report no memory per zone, so a
    # Guaranteed pod (last one) cannot be aligned anywhere
    pods = O.build_pod_objects(hdr, res, [
        {"containers": [O.container({"cpu": "1", "example.com/deviceA": "1"})]},
        {"containers": [O.container({"cpu": "1", "example.com/deviceA": "3"})]},
        {"containers": [O.container({"cpu": "1", "example.com/deviceB": "3"})]},
        {"containers": [O.container({"cpu": "2", "memory": "1Gi"}, {"cpu": "2", "memory": "1Gi"})]}])
    with NrtIngest(names, [res.names[i] for i in sorted(res.names) if i >= 8]) as ing:
        ing.feed(json.dumps(docs).encode())
        got_t = ing.nrt_objects()
        params = O.nrt_params(hdr, res, "LeastAllocated")
        a = oracle.Snapshot(nodes, pods, rc=res.table(hdr), nrt=want_t, nrt_params=params)
        b = oracle.Snapshot(nodes, pods, rc=ing.resource_classes(), nrt=got_t, nrt_params=params)
        fa, fb = a.filter_rows(3), b.filter_rows(3)
        assert fa.tolist() == fb.tolist()
        assert a.score_rows(3)[0].tolist() == b.score_rows(3)[0].tolist()
        # worker-node-a offers deviceA 1|2 and deviceB 2|1 per zone, worker-node-b deviceA 3 (node-0) and deviceB 3 (node-1)
        assert fa.tolist() == [[0, 0], [4, 0], [4, 0], [4, 4]]


def test_throughput(hdr):
    """50k nodes x 8 zones x 4 resources of CR JSON: decode rate on one core (reported, loosely bounded)"""
    n = 20_000
    names = [f"n{i}" for i in range(n)]
    zones = [{"name": f"node-{z}", "type": "Node",
              "resources": [{"name": r, "capacity": "64", "allocatable": "62", "available": f"{30 + z}"} for r in ("cpu", "memory", "hugepages-2Mi", "example.com/gpu")],
              "costs": [{"name": f"node-{k}", "value": 10 + abs(k - z)} for k in range(8)]} for z in range(8)]
    items = [{"apiVersion": "topology.node.k8s.io/v1alpha2", "kind": "NodeResourceTopology", "metadata": {"name": nm},
              "topologyPolicies": ["SingleNUMANodeContainerLevel"], "zones": zones} for nm in names]
    blob = json.dumps({"items": items}).encode()
    with NrtIngest(names) as ing:
        t0 = time.perf_counter()
        assert ing.feed(blob) == (n, 0)
        dt = time.perf_counter() - t0
        t = ing.nrt_objects().struct
        assert t.zone_ptr[n] == 8 * n and t.zres_ptr[8 * n] == 32 * n
    rate = len(blob) / dt / 1e6
    print(f"NRT JSON ingest: {len(blob) / 1e6:.1f} MB in {dt * 1e3:.0f} ms = {rate:.0f} MB/s, {n / dt:.0f} objects/s")
    assert rate > 20


def test_decoder_survives_mutated_input():
    """robustness: truncations, byte flips and spliced fragments of valid documents must yield a clean error or a decode,
    never a crash (the decoder reads caller-supplied bytes)"""
    docs = json.loads((GOLD / "nrt_manifests.json").read_text())
    base = json.dumps({"items": docs}).encode()
    rng = np.random.default_rng(99)
    outcomes = {"ok": 0, "err": 0}
    with NrtIngest([d["metadata"]["name"] for d in docs]) as ing:
        for it in range(4000):
            b = bytearray(base)
            kind = it % 4
            if kind == 0:
                b = b[: int(rng.integers(0, len(b)))]
            elif kind == 1:
                for _ in range(int(rng.integers(1, 6))):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            elif kind == 2:
                i, j = sorted(int(x) for x in rng.integers(0, len(b), 2))
                b = b[:i] + b[j:]
            else:
                i, j = sorted(int(x) for x in rng.integers(0, len(b), 2))
                b = b[:j] + b[i:j] + b[j:]
            try:
                ing.feed(bytes(b))
                outcomes["ok"] += 1
            except ValueError:
                outcomes["err"] += 1
        ing.feed(base)   # and the handle still works afterwards
        assert column(ing.nrt_objects().struct, "has_nrt", 2) == [1, 1]
    assert outcomes["err"] > 1000 and outcomes["ok"] > 0
