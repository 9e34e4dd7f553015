"""The placements the reference's integration tests expect (tests/golden/integration.py): on CPU the oracle's
one-pod-at-a-time cycle must produce them; on the GPU spx_commit_sequential / spx_eval_best must."""
import numpy as np
import pytest

from golden import integration as G
from helpers import ALLOCATABLE, LVRB, TLP, lvrb_params, tlp_params
from scheduler_plugins_amd import objects as O


def _tables(hdr, case):
    res = O.Resources()
    nodes = O.build_node_objects(hdr, res, [O.node(n["allocatable"], n["capacity"]) for n in case["nodes"]])
    pods = [O.pod([O.container(r)]) for r in case["pods"]]
    metrics = O.build_metrics_objects(hdr, len(case["nodes"]), case["metrics"], window_end=case["window_end"])
    return res, nodes, pods, metrics


def _oracle_sequence(hdr, oracle, case, plugin):
    res, nodes, pods, metrics = _tables(hdr, case)
    pod_t = O.build_pod_objects(hdr, res, pods)
    bound, placed = {}, []
    for i in range(len(pods)):
        assigned = O.build_assigned_objects(hdr, res, len(case["nodes"]), bound)
        osnap = oracle.Snapshot(nodes, pod_t, rc=res.table(hdr), metrics=metrics, assigned=assigned, tlp_params=tlp_params(hdr),
                                lvrb_params=lvrb_params(hdr))
        row = osnap.score_rows(plugin, i, i + 1, want_norm=False)[0][0]
        n = int(np.flatnonzero(row == row.max())[0])
        placed.append(case["nodes"][n]["name"])
        bound.setdefault(n, []).append((case["window_end"] + 1, pods[i]))  # bound after the metrics window closed
    return placed


@pytest.mark.parametrize("name,plugin", [("TLP", TLP), ("LVRB", LVRB)])
def test_trimaran_integration_placement_oracle(hdr, oracle, name, plugin):
    case = getattr(G, name)
    assert _oracle_sequence(hdr, oracle, case, plugin) == case["expected"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,plugin", [("TLP", TLP), ("LVRB", LVRB)])
def test_trimaran_integration_placement_gpu(gpu_required, hdr, name, plugin):
    from scheduler_plugins_amd.engine import Engine, mask_of
    case = getattr(G, name)
    res, nodes, pods, metrics = _tables(hdr, case)
    with Engine(0) as e:
        e.load_trimaran_objects(nodes, res.table(hdr), O.build_pod_objects(hdr, res, pods), metrics,
                                O.build_assigned_objects(hdr, res, len(case["nodes"]), {}))
        node, _, _, _ = e.commit_sequential(mask_of(plugin))
    assert [case["nodes"][int(n)]["name"] for n in node] == case["expected"]


def _alloc_setup(hdr, mode):
    case = G.ALLOCATABLE
    res = O.Resources()
    nodes = O.build_node_objects(hdr, res, [O.node(n["allocatable"], n["capacity"]) for n in case["nodes"]])
    pods = O.build_pod_objects(hdr, res, [O.pod([O.container(r)]) for _, r in case["pods"]])
    # NodeResourcesFit stand-in: a pod is feasible where its memory request fits the node's allocatable memory
    feas = np.array([[O.parse_quantity(r["memory"]) <= O.parse_quantity(n["allocatable"]["memory"]) for n in case["nodes"]]
                     for _, r in case["pods"]], dtype=np.uint8)
    weights = {res.id(k): v for k, v in case["weights"].items()}
    return case, res, nodes, pods, feas, weights


@pytest.mark.parametrize("mode", ["Least", "Most"])
def test_allocatable_integration_placement_oracle(hdr, oracle, mode):
    case, res, nodes, pods, feas, weights = _alloc_setup(hdr, mode)
    from scheduler_plugins_amd._abi import Table
    params = Table(hdr, "spx_allocatable_params", mode={"Least": 0, "Most": 1}[mode], n_res=len(weights),
                   res=np.array(list(weights.keys()), dtype=np.int32), weight=np.array(list(weights.values()), dtype=np.int64))
    osnap = oracle.Snapshot(nodes, pods, rc=res.table(hdr), alloc_params=params)
    _, norm = osnap.score_rows(ALLOCATABLE, mask=feas)
    for i, (name, _) in enumerate(case["pods"]):
        row = np.where(feas[i] != 0, norm[i], -1)
        got = {case["nodes"][int(n)]["name"] for n in np.flatnonzero(row == row.max())}
        assert got == case["expected"][mode][name], (name, got)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["Least", "Most"])
def test_allocatable_integration_placement_gpu(gpu_required, hdr, mode):
    from scheduler_plugins_amd.engine import Engine, mask_of
    case, res, nodes, pods, feas, weights = _alloc_setup(hdr, mode)
    with Engine(0) as e:
        e.set_allocatable(mode, weights)
        e.upload_alloc_nodes(e.flatten_alloc_nodes(nodes, res.table(hdr)))
        e.n_pods = len(case["pods"])
        e.upload_feasible_mask(feas)
        e.eval(mask_of(ALLOCATABLE))
        e.sync()
        for i, (name, _) in enumerate(case["pods"]):
            row = e.scores(ALLOCATABLE, i).astype(np.int64)
            row = np.where(feas[i] != 0, row, -1)
            got = {case["nodes"][int(n)]["name"] for n in np.flatnonzero(row == row.max())}
            assert got == case["expected"][mode][name], (name, got)


def test_lroc_integration_placement_oracle(hdr, oracle):
    """test/integration/lowriskovercommitment_test.go: pod-3 lands on node-1 (scores 100 vs 88)"""
    from golden.integration import LROC as CASE
    from helpers import LROC, lroc_params

    res = O.Resources()
    nodes = O.build_node_objects(hdr, res, [O.node(n["allocatable"], n["capacity"]) for n in CASE["nodes"]])
    pods = O.build_pod_objects(hdr, res, [{"containers": [O.container(*CASE["pod"])]}])
    node_pods = O.build_node_pods_objects(hdr, res, len(CASE["nodes"]), {i: [{"containers": [O.container(r, l)]} for r, l in ps] for i, ps in CASE["on_node"].items()})
    snap = oracle.Snapshot(nodes, pods, metrics=O.build_metrics_objects(hdr, len(CASE["nodes"]), CASE["metrics"]), node_pods=node_pods,
                           lroc_params=lroc_params(hdr, **CASE["params"]))
    raw, _ = snap.score_rows(LROC)
    assert raw[0].tolist() == CASE["scores"]
    assert CASE["nodes"][int(raw[0].argmax())]["name"] == CASE["expected"]


def test_peaks_integration_placement_oracle(hdr, oracle):
    """test/integration/peaks_test.go: pod-1 -> node-1, pod-2 -> node-2 (node-1 is full for it)"""
    import numpy as np

    from golden.integration import PEAKS as CASE
    from helpers import PEAKS, power_models

    res = O.Resources()
    nodes = O.build_node_objects(hdr, res, [O.node(n["allocatable"], n["capacity"]) for n in CASE["nodes"]])
    pods = O.build_pod_objects(hdr, res, [{"containers": [O.container(p)]} for p in CASE["pods"]])
    snap = oracle.Snapshot(nodes, pods, metrics=O.build_metrics_objects(hdr, 3, CASE["metrics"]), power_models=power_models(hdr, CASE["models"]))
    mask = np.array(CASE["feasible"], dtype=np.uint8)
    _, norm = snap.score_rows(PEAKS, mask=mask)
    for i, want in enumerate(CASE["expected"]):
        row = np.where(mask[i] != 0, norm[i], -1)
        assert CASE["nodes"][int(row.argmax())]["name"] == want and (row == row.max()).sum() == 1
