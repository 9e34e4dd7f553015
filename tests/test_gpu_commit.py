"""spx_commit_sequential: pods scheduled one after the other, each seeing the commits before it.
The check rebuilds, with the CPU oracle, what upstream's one-pod-at-a-time cycle computes: after every decision the bound
pod joins trimaran's ScheduledPodsCache image (objects.build_assigned_objects) and the next pod's row is scored against
that state."""
import numpy as np
import pytest

from helpers import ALLOCATABLE, LVRB, TLP, lvrb_params, tlp_params
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd.engine import Engine, mask_of

pytestmark = pytest.mark.gpu
WINDOW_END = 1_700_000_000


def _scenario(hdr, n_nodes, n_pods, seed):
    rng = np.random.default_rng(seed)
    res = O.Resources()
    nodes, metrics = [], {}
    for i in range(n_nodes):
        cpu = int(rng.choice([4, 8, 16, 32]))
        nodes.append(O.node({"cpu": f"{cpu}", "memory": f"{int(rng.choice([16, 32, 64]))}Gi"}, {"cpu": f"{cpu}", "memory": "64Gi"}))
        if rng.random() < 0.9:
            metrics[i] = [("CPU", "AVG", float(rng.integers(5, 70))), ("CPU", "STD", float(rng.integers(0, 20))),
                          ("Memory", "AVG", float(rng.integers(5, 80))), ("Memory", "STD", float(rng.integers(0, 20)))]
    pods = []
    for _ in range(n_pods):
        cpu = int(rng.choice([250, 500, 1000, 2000]))
        req = {"cpu": f"{cpu}m", "memory": f"{int(rng.choice([128, 512, 2048]))}Mi"}
        pods.append(O.pod([O.container(req, req if rng.random() < 0.5 else None)]))
    earlier = {int(n): [(WINDOW_END + int(rng.integers(-200, 50)), pods[int(rng.integers(0, n_pods))])] for n in rng.choice(n_nodes, n_nodes // 5, replace=False)}
    return res, nodes, metrics, pods, earlier


@pytest.mark.parametrize("plugins,weights", [((ALLOCATABLE, TLP), {ALLOCATABLE: 1, TLP: 1}), ((TLP,), {TLP: 1}),
                                             ((ALLOCATABLE, TLP, LVRB), {ALLOCATABLE: 1, TLP: 3, LVRB: 2})])
@pytest.mark.parametrize("n_nodes,n_pods,seed", [(23, 90, 1), (70, 60, 2), (1100, 40, 3)])
@pytest.mark.parametrize("state", ["registers", "memory"])
def test_commit_sequential_matches_one_pod_at_a_time(gpu_required, hdr, oracle, state, plugins, weights, n_nodes, n_pods, seed):
    """both variants of the loop: node state resident in registers (up to 10240 nodes) and re-read from memory per pod"""
    res, nodes, metrics, pods, earlier = _scenario(hdr, n_nodes, n_pods, seed)
    node_t = O.build_node_objects(hdr, res, nodes)
    pod_t = O.build_pod_objects(hdr, res, pods)
    met_t = O.build_metrics_objects(hdr, n_nodes, metrics, window_end=WINDOW_END)
    rc = res.table(hdr)
    with Engine(0) as e:
        e.set_option("COMMIT_FROM_MEMORY", 1 if state == "memory" else 0)
        e.load_trimaran_objects(node_t, rc, pod_t, met_t, O.build_assigned_objects(hdr, res, n_nodes, earlier))
        e.set_plugin_weights(weights)
        got_node, got_score, got_ties, got_missing = e.commit_sequential(mask_of(*plugins))
        alloc_params = e.alloc_params
    # one pod at a time with the oracle
    bound = {n: list(v) for n, v in earlier.items()}
    for i in range(n_pods):
        assigned = O.build_assigned_objects(hdr, res, n_nodes, bound)
        osnap = oracle.Snapshot(node_t, pod_t, rc=rc, metrics=met_t, assigned=assigned, alloc_params=alloc_params,
                                tlp_params=tlp_params(hdr), lvrb_params=lvrb_params(hdr))
        total = np.zeros(n_nodes, np.int64)
        for p in plugins:
            raw, norm = osnap.score_rows(p, i, i + 1, want_norm=(p == ALLOCATABLE))
            row = norm[0] if p == ALLOCATABLE else raw[0]
            total += weights[p] * row.astype(np.int64).clip(0, 255)
        best = int(total.max())
        tie_set = np.flatnonzero(total == best)
        assert got_score[i] == best, (i, got_score[i], best)
        assert got_node[i] == tie_set[0] and got_ties[i] == tie_set.size, (i, got_node[i], tie_set)
        bound.setdefault(int(got_node[i]), []).append((WINDOW_END + 1, pods[i]))  # bound "now": after the metrics window
    assert len(set(got_node.tolist())) > 1  # the commits moved the decision around


def test_commit_sequential_rejects_filter_plugins(gpu_required, hdr):
    from helpers import NRT
    from scheduler_plugins_amd import synth
    snap = synth.trimaran_snapshot(hdr, 10, 5)
    with Engine(0) as e:
        e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        with pytest.raises(Exception):
            e.commit_sequential(mask_of(TLP, NRT))
