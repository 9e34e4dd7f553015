"""spx_commit_sequential: pods scheduled one after the other, each seeing the commits before it.
The check rebuilds, with the CPU oracle, what upstream's one-pod-at-a-time cycle computes: after every decision the bound
pod joins trimaran's ScheduledPodsCache image (objects.build_assigned_objects) and the next pod's row is scored against
that state."""
import numpy as np
import pytest

from helpers import ALLOCATABLE, LVRB, NRT, TLP, lvrb_params, tlp_params
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


@pytest.mark.parametrize("cores,n_pods", [(4000, 700), (9000, 60), (2, 300)])
def test_commit_sequential_absurd_pod_values(gpu_required, hdr, cores, n_pods):
    """the register-resident loop keeps the millicores it committed per node in 32-bit LDS cells: unchecked adds when the batch's
    values are float32 integers summing below 2^31 (checked once per launch), checked adds folded into the column otherwise
    (700 x 4000 cores = 2.8e9 millicores), and values of 2^23 millicores or more take the float64 sequence for the whole row
    (9000 cores) — all three against the from-memory float64 kernel"""
    rng = np.random.default_rng(cores)
    res = O.Resources()
    n_nodes = 50
    nodes = [O.node({"cpu": "64", "memory": "256Gi"}, {"cpu": "64", "memory": "256Gi"}) for _ in range(n_nodes)]
    metrics = {i: [("CPU", "AVG", float(rng.integers(5, 70))), ("CPU", "STD", 3.0)] for i in range(n_nodes)}
    pods = [O.pod([O.container({"cpu": f"{cores + int(rng.integers(0, 3))}"})]) for _ in range(n_pods)]
    node_t, pod_t, rc = O.build_node_objects(hdr, res, nodes), O.build_pod_objects(hdr, res, pods), res.table(hdr)
    met_t = O.build_metrics_objects(hdr, n_nodes, metrics, window_end=WINDOW_END)
    out = {}
    with Engine(0) as e:
        e.load_trimaran_objects(node_t, rc, pod_t, met_t, O.build_assigned_objects(hdr, res, n_nodes, {}))
        for state in ("registers", "memory"):
            e.set_option("COMMIT_FROM_MEMORY", 1 if state == "memory" else 0)
            out[state] = e.commit_sequential(mask_of(ALLOCATABLE, TLP), want_ties=True)
            assert e.commit_path() == 1
    for a, b in zip(out["registers"], out["memory"]):
        assert np.array_equal(a, b)
    assert out["registers"][3].sum() > 0  # the committed millicores reached the missing-utilisation column


def test_commit_sequential_rejects_unknown_plugins(gpu_required, hdr):
    from helpers import LROC
    from scheduler_plugins_amd import synth
    snap = synth.trimaran_snapshot(hdr, 10, 5)
    with Engine(0) as e:
        e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        with pytest.raises(Exception, match="supports"):
            e.commit_sequential(mask_of(TLP, LROC))
        with pytest.raises(Exception, match="NRT slot/node/pod tables"):
            e.commit_sequential(mask_of(TLP, NRT))   # a Filter plugin whose tables are missing


# ------------------------------------------------------------------ the full profile, one pod at a time
REGIONS = {"r0": ["z0", "z1"], "r1": ["z2", "z3"]}
REGION_COSTS = {"r0": [("r1", 20)], "r1": [("r0", 20)]}
ZONE_COSTS = {"z0": [("z1", 5)], "z1": [("z0", 5)], "z2": [("z3", 7)], "z3": [("z2", 7)]}
GROUPS = [
    {"name": "g0", "workloads": [{"selector": "a", "dependencies": [("b", 6), ("c", 25)]}, {"selector": "b", "dependencies": [("c", 4)]},
                                  {"selector": "c", "dependencies": []}],
     "topology_order": [("a", 1), ("b", 2), ("c", 3)]},
    {"name": "g1", "workloads": [{"selector": "a", "dependencies": [("d", 1)]}, {"selector": "d", "dependencies": [("a", 30)]}],
     "topology_order": [("a", 2), ("d", 1)]},
]


def _full_scenario(hdr, n_nodes, n_pods, seed):
    rng = np.random.default_rng(seed)
    zone_of = [z for r in REGIONS.values() for z in r]
    nrts, nodes, node_labels = [], [], []
    for i in range(n_nodes):
        nz = int(rng.choice([2, 4]))
        zones = []
        for z in range(nz):
            rl = {"cpu": f"{int(rng.integers(2, 9))}", "memory": f"{int(rng.integers(2, 17))}Gi"}
            if rng.random() < 0.6:
                rl["vendor.io/gpu"] = str(int(rng.integers(0, 4)))
            zones.append({"name": f"node-{z}", "type": "Node", "resources": rl,
                          "costs": {f"node-{o}": (10 if o == z else 20) for o in range(nz)}})
        policy = "SingleNUMANodePodLevel" if rng.random() < 0.35 else "SingleNUMANodeContainerLevel"
        nrts.append(None if rng.random() < 0.05 else O.nrt(zones, [policy]))
        zn = zone_of[int(rng.integers(0, 4))]
        rg = [r for r, zs in REGIONS.items() if zn in zs][0]
        node_labels.append((rg, zn))
        d = O.node_from_zones(zones)
        d["capacity"] = {"cpu": "64", "memory": "256Gi"}
        nodes.append(d)
    pods, meta = [], []
    for i in range(n_pods):
        ctrs = []
        for _ in range(int(rng.integers(1, 3))):
            req = {"cpu": str(int(rng.integers(1, 4))), "memory": f"{int(rng.integers(1, 5))}Gi"}
            if rng.random() < 0.2:
                req["vendor.io/gpu"] = "1"
            ctrs.append(O.container(req, dict(req)))   # Guaranteed
        g = int(rng.integers(-1, 2))
        sel = None if g < 0 else str(rng.choice(["a", "b", "c"] if g == 0 else ["a", "d"]))
        meta.append((g, sel))
        pods.append(dict(containers=ctrs, priority=int(rng.choice([0, 10, 100])), ns=int(rng.integers(0, 3)), queue_ts=i))
    metrics = {i: [("CPU", "AVG", float(rng.integers(5, 60))), ("CPU", "STD", float(rng.integers(0, 10))),
                   ("Memory", "AVG", float(rng.integers(5, 60))), ("Memory", "STD", float(rng.integers(0, 10)))] for i in range(n_nodes) if rng.random() < 0.9}
    quotas = [{"min": {"cpu": "20", "memory": "40Gi"}, "max": {"cpu": "40", "memory": "120Gi", "vendor.io/gpu": "3"}, "used": {"cpu": "6", "memory": "10Gi"}},
              None,
              {"min": {"cpu": "10", "memory": "200Gi", "vendor.io/gpu": "2"}, "max": {"cpu": "24", "memory": "300Gi"}, "used": {"cpu": "2", "memory": "1Gi"}}]
    nominated = sorted(int(x) for x in rng.choice(n_pods, max(1, n_pods // 6), replace=False))
    return nrts, nodes, node_labels, pods, meta, metrics, quotas, nominated


@pytest.mark.parametrize("kernels", ["coop", "coop-most", "graph", "reference", "fast-direct"])
@pytest.mark.parametrize("n_nodes,n_pods,seed", [(40, 70, 1), (150, 60, 2), (700, 50, 3)])
def test_commit_sequential_full_profile(gpu_required, hdr, oracle, kernels, n_nodes, n_pods, seed):
    """NRT + NetworkOverhead + CapacityScheduling + Allocatable + TLP + LVRB scheduled one pod at a time.  After every decision the
    test applies what the reference's Reserve hooks do to its own Python-side state (NRT assumed resources on the node, the
    AppGroup's scheduled list, the namespace's Used, the nominated-pod list, trimaran's ScheduledPodsCache), rebuilds the object
    tables from it and lets the CPU oracle evaluate the next pod's row from scratch; node, weighted score, tie-set size and the
    unschedulable verdicts must equal the device loop's.  Forms of the loop: the cooperative persistent kernel (one workgroup per 256
    nodes: 700 nodes = three workgroups exchanging granules), the per-pod launches replayed from a graph, the same with the
    reference-arithmetic kernels, and plain per-pod launches."""
    from helpers import CAPACITY, NETOVERHEAD, NRT
    nrts, nodes, node_labels, pods, meta, metrics, quotas, nominated = _full_scenario(hdr, n_nodes, n_pods, seed)
    res = O.Resources()
    res.id("vendor.io/gpu")
    regions, zones = O.Interner(), O.Interner()
    nt_t = O.build_nettopo_objects(hdr, regions, zones, REGION_COSTS, ZONE_COSTS)
    for i, (rg, zn) in enumerate(node_labels):
        nodes[i]["region"], nodes[i]["zone"] = regions.id(rg), zones.id(zn)
    sel = O.Interner(["a", "b", "c", "d"])
    sel.freeze_sorted()
    pod_dicts = [O.pod(p["containers"], priority=p["priority"], queue_ts=p["queue_ts"], ns=p["ns"], appgroup=g, selector=sel.id(s))
                 for p, (g, s) in zip(pods, meta)]
    node_t = O.build_node_objects(hdr, res, nodes)
    pod_t = O.build_pod_objects(hdr, res, pod_dicts)
    met_t = O.build_metrics_objects(hdr, n_nodes, metrics, window_end=WINDOW_END)
    rc = res.table(hdr)
    if n_nodes > 200 and kernels in ("reference", "fast-direct"):
        pytest.skip("the large scenario exists for the cooperative kernel's granule exchange")
    params = O.nrt_params(hdr, res, "MostAllocated" if kernels == "coop-most" else "LeastAllocated")
    names = {f"n{i}": i for i in range(n_nodes)}

    def tables(assumed, placed, used, nom):
        nrt_t = O.build_nrt_objects(hdr, res, nrts, assumed=assumed)
        ag_t = O.build_appgroup_objects(hdr, sel, [dict(g, placed=[(s, f"n{n}") for s, n in placed[gi]]) for gi, g in enumerate(GROUPS)], names)
        q = [None if qq is None else dict(qq, used=used[k]) for k, qq in enumerate(quotas)]
        quota_t = O.build_quota_objects(hdr, res, q, nominated=[(pods[j]["ns"], pods[j]["priority"], j, pod_dicts[j]) for j in nom])
        return nrt_t, ag_t, quota_t

    used0 = [None if q is None else q["used"] for q in quotas]
    nrt_t, ag_t, quota_t = tables({}, [[], []], used0, nominated)
    weights = {ALLOCATABLE: 1, TLP: 2, LVRB: 1, NRT: 3, NETOVERHEAD: 2}
    plugins = (ALLOCATABLE, TLP, LVRB, NRT, NETOVERHEAD, CAPACITY)
    with Engine(0) as e:
        if kernels == "reference":
            e.force_reference_kernels(TLP, LVRB, NRT, NETOVERHEAD)
        if kernels == "fast-direct":   # plain launches per pod instead of the replayed graph (rows from the host, not the device counter)
            e.set_option("COMMIT_FROM_MEMORY", 1)
        if kernels == "graph":
            e.set_option("COMMIT_COOP", 0)
        e.load_trimaran_objects(node_t, rc, pod_t, met_t, O.build_assigned_objects(hdr, res, n_nodes, {}))
        e.load_nrt_objects(node_t, nrt_t, rc, pod_t, params)
        e.load_network_objects(node_t, pod_t, ag_t, nt_t)
        e.load_quota_objects(pod_t, rc, quota_t)
        e.set_plugin_weights(weights)
        assert e.kernel_path(NRT) == (0 if kernels == "reference" else 1)
        got_node, got_score, got_ties, _ = e.commit_sequential(mask_of(*plugins))
        assert e.commit_path() == (3 if kernels.startswith("coop") else 2)
        # the snapshot is intact afterwards: a frozen-snapshot evaluation gives what a fresh engine gives
        e.eval(mask_of(*plugins))
        e.sync()
        frozen_nrt = e.all_status(NRT)
        alloc_params = e.alloc_params
        pod_req = e.nrt_soa["pods"]["pod_req"].reshape(n_pods, -1)
        pod_present = e.nrt_soa["pods"]["pod_present"]
        slot_res = e.nrt_soa["slots"].array("slot_res")
        qcols = e.flatten_quota(pod_t, rc, quota_t)["cols"]
    res_name = {v: k for k, v in res.ids.items()}
    scalar_names = [res_name[int(r)] for r in quota_t.array("scalar_res")[: quota_t.struct.n_scalar_slots]]

    def effective_request(i):  # GetPodEffectiveRequest as a resource list (the reserve store's entry)
        rl = {}
        for s in range(pod_req.shape[1]):
            if (pod_present[i] >> s) & 1:
                name = res_name[int(slot_res[s])]
                rl[name] = f"{int(pod_req[i, s])}m" if name == "cpu" else int(pod_req[i, s])
        return rl

    def add_used(u, i):  # reserveResource elasticquota.go:89-98, on framework.Resource fields
        v = qcols["pod_req"][i * 8:(i + 1) * 8]
        base, _ = O._resource_vec(res, [res.ids[n] for n in scalar_names], u)
        out = {"MilliCPU": base[0] + int(v[0]), "Memory": base[1] + int(v[1]), "EphemeralStorage": base[2] + int(v[2]), "AllowedPodNumber": base[3] + int(v[3]),
               "ScalarResources": {}}
        keys = dict((u or {}).get("ScalarResources", {})) if u and "ScalarResources" in u else {k: None for k in (u or {}) if O.is_scalar_resource_name(k)}
        for si, name in enumerate(scalar_names):
            if name in keys or (qcols["pod_req_present"][i] >> (4 + si)) & 1:
                out["ScalarResources"][name] = base[4 + si] + int(v[4 + si])
        return out

    assumed, placed, used, nom, bound = {}, [[], []], list(used0), list(nominated), {}
    n_unsched = 0
    with Engine(0) as fresh:   # (the frozen NRT status the first engine reported after the loop must be the untouched snapshot's)
        fresh.load_nrt_objects(node_t, nrt_t, rc, pod_t, params)
        fresh.eval(mask_of(NRT))
        fresh.sync()
        assert np.array_equal(fresh.all_status(NRT), frozen_nrt)
    for i in range(n_pods):
        nrt_i, ag_i, quota_i = tables(assumed, placed, used, nom)
        osnap = oracle.Snapshot(node_t, pod_t, rc=rc, metrics=met_t, assigned=O.build_assigned_objects(hdr, res, n_nodes, bound), alloc_params=alloc_params,
                                tlp_params=tlp_params(hdr), lvrb_params=lvrb_params(hdr), nrt=nrt_i, nrt_params=params, appgroups=ag_i, nettopo=nt_t)
        pre = oracle.lib().orc_capacity_prefilter(pod_t.ref(), rc.ref(), quota_i.ref(), i)
        nrt_st = osnap.filter_rows(NRT, i, i + 1)[0]
        net_st = osnap.filter_rows(NETOVERHEAD, i, i + 1)[0]
        feasible = (nrt_st == 0) & (net_st == 0)
        full = lambda m: np.concatenate([np.zeros((i, n_nodes), np.uint8), m[None, :].astype(np.uint8)])
        total = np.zeros(n_nodes, np.int64)
        for p in (TLP, LVRB, NRT):
            total += weights[p] * osnap.score_rows(p, i, i + 1, want_norm=False)[0][0].clip(0, 255)
        total += weights[NETOVERHEAD] * osnap.score_rows(NETOVERHEAD, i, i + 1, mask=full(nrt_st == 0), want_raw=False)[1][0]
        total += weights[ALLOCATABLE] * osnap.score_rows(ALLOCATABLE, i, i + 1, mask=full(feasible), want_raw=False)[1][0]
        if pre != 0 or not feasible.any():
            assert got_node[i] == -1 and got_ties[i] == 0, (i, pre, got_node[i])
            n_unsched += 1
            continue
        best = int(total[feasible].max())
        tie_set = np.flatnonzero(feasible & (total == best))
        assert (got_node[i], got_score[i], got_ties[i]) == (tie_set[0], best, tie_set.size), (i, got_node[i], got_score[i], got_ties[i], tie_set[:4], best)
        n = int(got_node[i])
        if nrts[n] is not None:
            assumed.setdefault(n, []).append(effective_request(i))
        g, s = meta[i]
        if g >= 0:
            placed[g].append((s, n))
        k = pods[i]["ns"]
        if quotas[k] is not None:
            used[k] = add_used(used[k], i)
        nom = [j for j in nom if j != i]
        bound.setdefault(n, []).append((WINDOW_END + 1, pod_dicts[i]))
    assert 0 < n_unsched < n_pods and len(set(got_node.tolist())) > 3
