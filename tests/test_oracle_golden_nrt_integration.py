"""The reference's NodeResourceTopologyMatch integration table (test/integration/noderesourcetopology_test.go, 29 cases,
transcribed to tests/golden/nrt_integration.json): one pod, two nodes, the scoring strategy of the profile the pod
names, and the nodes it may land on (empty = must stay pending).  Upstream's other default plugins tie on the two
identical, empty nodes, so Filter + Score of this plugin decide: every node with the best score among those that pass
the Filter must be an allowed one."""
import json
from pathlib import Path

import numpy as np
import pytest

from helpers import NRT
from scheduler_plugins_amd import objects as O

G = json.loads((Path(__file__).resolve().parent / "golden" / "nrt_integration.json").read_text())


def build(hdr, case):
    res = O.Resources()
    names = G["node_names"]
    nodes = O.build_node_objects(hdr, res, [O.node(G["node_capacity"], G["node_capacity"]) for _ in names])
    by_name = {n["name"]: n for n in case["nrts"]}
    nrts = O.build_nrt_objects(hdr, res, [O.nrt([{**z, "resources": [tuple(r) for r in z["resources"]]} for z in by_name[n]["zones"]],
                                               by_name[n]["policies"], by_name[n]["attributes"]) if n in by_name else None for n in names])
    p = case["pod"]
    pods = O.build_pod_objects(hdr, res, [O.pod([O.container(c.get("requests"), c.get("limits")) for c in p["containers"]],
                                                [O.container(c.get("requests"), c.get("limits")) for c in p["init_containers"]])])
    return res, nodes, nrts, pods, O.nrt_params(hdr, res, case["strategy"])


def check(case, status, score):
    feasible = np.flatnonzero(status == 0)
    if not case["expected_nodes"]:
        assert feasible.size == 0, ("pod must stay pending", status)
        return
    assert feasible.size > 0, status
    best = score[feasible].max()
    winners = {G["node_names"][int(n)] for n in feasible if score[n] == best}
    assert winners <= set(case["expected_nodes"]), (winners, case["expected_nodes"], status, score)


@pytest.mark.parametrize("case", G["cases"], ids=lambda c: f"L{c['line']}")
def test_nrt_integration_oracle(hdr, oracle, case):
    res, nodes, nrts, pods, params = build(hdr, case)
    osnap = oracle.Snapshot(nodes, pods, rc=res.table(hdr), nrt=nrts, nrt_params=params)
    check(case, osnap.filter_rows(NRT)[0], osnap.score_rows(NRT, want_norm=False)[0][0])


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["float64", "generic"])
@pytest.mark.parametrize("case", G["cases"], ids=lambda c: f"L{c['line']}")
def test_nrt_integration_gpu(gpu_required, hdr, case, kernel):
    from scheduler_plugins_amd.engine import Engine, mask_of
    res, nodes, nrts, pods, params = build(hdr, case)
    with Engine(0) as e:
        if kernel == "generic":
            e.force_reference_kernels(NRT)
        e.load_nrt_objects(nodes, nrts, res.table(hdr), pods, params)
        e.eval(mask_of(NRT))
        e.sync()
        check(case, e.status(NRT, 0), e.raw(NRT, 0))
