"""The reference's NRT cache integration table (test/integration/noderesourcetopology_cache_test.go:111-644,
TestTopologyCachePluginWithoutUpdates, transcribed to tests/golden/nrt_cache_integration.json): pods created one after the other
against two nodes whose NRT objects never change.  With the OverReserve cache (overreserve.go:170-203) a bound pod's request is
charged to EVERY zone of its node until a resync, and deletes are ignored; with DiscardReserved nothing is charged once PostBind
ran.  Replayed on the CPU with the oracle (Filter + LeastAllocated Score per pod, the assumed-pod list growing as pods bind) and on
the GPU with the device's own commit loop (spx_commit_sequential with NodeResourceTopologyMatch in the mask = the OverReserve
bookkeeping, kernels_commit*.hip) — SURVEY 8a N11."""
import json
from pathlib import Path

import numpy as np
import pytest

from helpers import NRT
from scheduler_plugins_amd import objects as O

G = json.loads((Path(__file__).resolve().parent / "golden" / "nrt_cache_integration.json").read_text())


def _nodes_and_nrts(case):
    """createNodesFromNodeResourceTopologies (nrtutils.go:73-90): capacity = the zones' capacities summed per resource, pods 128"""
    nodes, nrts, names = [], [], []
    for n in case["nrts"]:
        cap = {}
        for z in n["zones"]:
            for name, capacity, _available in z["resources"]:
                cap[name] = cap.get(name, 0) + O.parse_quantity(capacity)
        rl = {k: (f"{int(v * 1000)}m" if k == "cpu" else str(int(v))) for k, v in cap.items()}
        rl.update(G["node_extra_capacity"])
        nodes.append(O.node(rl, rl))
        nrts.append(O.nrt([{**z, "resources": [tuple(r) for r in z["resources"]]} for z in n["zones"]], n["policies"], n["attributes"]))
        names.append(n["name"])
    return nodes, nrts, names


def _pods(case):
    return [s for s in case["steps"] if "pod" in s]


def _pod_dict(step):
    return O.pod([O.container(c.get("requests"), c.get("limits")) for c in step["containers"]])


def _effective_request(step):
    """util.GetPodEffectiveRequest of these pods (no init containers, no overhead): the containers' requests summed"""
    out = {}
    for c in step["containers"]:
        for k, v in c["requests"].items():
            out[k] = out.get(k, 0) + O.parse_quantity(v)
    return {k: (f"{int(v * 1000)}m" if k == "cpu" else int(v)) for k, v in out.items()}


def _check(step, names, feasible, score):
    exp = step["expected_node"]
    if exp == "":
        assert feasible.size == 0, ("pod must stay pending", step["pod"], feasible)
        return None
    assert feasible.size > 0, (step["pod"], "no node passes the Filter")
    best = score[feasible].max()
    winners = [int(n) for n in feasible if score[n] == best]
    if exp != "*":
        assert {names[n] for n in winners} == {exp}, (step["pod"], [names[n] for n in winners], exp)
    return winners[0]


@pytest.mark.parametrize("case", G["cases"], ids=lambda c: f"L{c['line']}")
def test_cache_integration_oracle(hdr, oracle, case):
    res = O.Resources()
    nodes, nrts, names = _nodes_and_nrts(case)
    node_t = O.build_node_objects(hdr, res, nodes)
    params = O.nrt_params(hdr, res, case["strategy"])
    assumed = {}
    for step in case["steps"]:
        if "delete" in step:
            continue  # OverReserve ignores deletes until a resync (overreserve.go:205-224); DiscardReserved holds nothing to release
        pod_t = O.build_pod_objects(hdr, res, [_pod_dict(step)])
        nrt_t = O.build_nrt_objects(hdr, res, nrts, assumed=assumed if case["cache"] == "OverReserve" else None)
        osnap = oracle.Snapshot(node_t, pod_t, rc=res.table(hdr), nrt=nrt_t, nrt_params=params)
        status = osnap.filter_rows(NRT)[0]
        score = osnap.score_rows(NRT, want_norm=False)[0][0]
        n = _check(step, names, np.flatnonzero(status == 0), score)
        if n is not None:
            assumed.setdefault(n, []).append(_effective_request(step))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["float64", "generic"])
@pytest.mark.parametrize("case", G["cases"], ids=lambda c: f"L{c['line']}")
def test_cache_integration_gpu(gpu_required, hdr, case, kernel):
    """OverReserve cases: the case's pods as ONE batch through spx_commit_sequential (the device charges every zone of the chosen
    node before the next pod's row is swept); DiscardReserved cases: the frozen snapshot's rows (nothing is charged)."""
    from scheduler_plugins_amd.engine import Engine, mask_of
    res = O.Resources()
    nodes, nrts, names = _nodes_and_nrts(case)
    steps = _pods(case)
    node_t = O.build_node_objects(hdr, res, nodes)
    pod_t = O.build_pod_objects(hdr, res, [_pod_dict(s) for s in steps])
    nrt_t = O.build_nrt_objects(hdr, res, nrts)
    with Engine(0) as e:
        if kernel == "generic":
            e.force_reference_kernels(NRT)
        e.load_nrt_objects(node_t, nrt_t, res.table(hdr), pod_t, O.nrt_params(hdr, res, case["strategy"]))
        e.set_plugin_weights({NRT: 1})
        if case["cache"] == "OverReserve":
            got, _, ties, _ = e.commit_sequential(mask_of(NRT))
        else:
            e.eval(mask_of(NRT))
            e.eval_best(mask_of(NRT))
            e.sync()
            got, _, ties, _ = e.best()
    for i, s in enumerate(steps):
        exp = s["expected_node"]
        if exp == "":
            assert got[i] == -1, (s["pod"], got[i])
        elif exp == "*":
            assert got[i] >= 0, s["pod"]
        else:
            assert got[i] >= 0 and names[int(got[i])] == exp and ties[i] == 1, (s["pod"], got[i], ties[i])
