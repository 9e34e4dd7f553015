"""GPU parity (through the C ABI) for networkaware NetworkOverhead (+ TopologicalSort keys)."""
import numpy as np
import pytest

from golden import network as GN
from helpers import NETOVERHEAD
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd import synth
from scheduler_plugins_amd.engine import Engine, mask_of
from test_oracle_golden_network import build

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", GN.SCORE_CASES, ids=lambda c: f"L{c['line']}")
def test_score_golden(gpu_required, hdr, case):
    nodes, pods, ag, nt = build(hdr, GN.SCORE_PLACED, [(case["appgroup"], case["selector"])])
    with Engine(0) as e:
        e.load_network_objects(nodes, pods, ag, nt)
        e.eval(mask_of(NETOVERHEAD))
        e.sync()
        assert e.raw(NETOVERHEAD, 0, 0).tolist() == case["before"]   # Score(): accumulated cost per node
        # the reference test normalises the full 8-node list; our table normalises over the nodes that pass the
        # plugin's own Filter (upstream semantics), which coincide when no node is filtered out
        st = e.status(NETOVERHEAD, 0)
        if not st.any():
            assert e.scores(NETOVERHEAD, 0).tolist() == case["after"]


@pytest.mark.parametrize("case", GN.FILTER_CASES, ids=lambda c: f"L{c['line']}")
def test_filter_golden(gpu_required, hdr, case):
    nodes, pods, ag, nt = build(hdr, GN.FILTER_PLACED, [(case["appgroup"], case["selector"])])
    with Engine(0) as e:
        e.load_network_objects(nodes, pods, ag, nt)
        e.eval(mask_of(NETOVERHEAD))
        e.sync()
        n = case["node"]
        assert e.status(NETOVERHEAD, 0)[n] == (1 if case["want"] else 0)
        if case["want"]:  # "... Satisfied: 0 Violated: 1"
            assert (e.raw(NETOVERHEAD, 0, 1)[n], e.raw(NETOVERHEAD, 0, 2)[n]) == case["want"]


@pytest.mark.parametrize("case", GN.LESS_CASES, ids=lambda c: f"L{c['line']}")
def test_toposort_less_golden(gpu_required, hdr, case):
    nodes, pods, ag, nt = build(hdr, [], [case["p1"], case["p2"]])
    with Engine(0) as e:
        e.load_network_objects(nodes, pods, ag, nt)
        assert bool(e.toposort_less(pods, [0], [1])[0]) == case["want"]


@pytest.mark.parametrize("case", GN.QUEUE_ORDER_CASES, ids=lambda c: f"L{c['line']}")
def test_toposort_queue_order_golden(gpu_required, hdr, case):
    """test/integration/topologicalsort_test.go:253-342 through the product's comparator (all pairs in one call)."""
    n = len(case["created"])
    nodes, pods, ag, nt = build(hdr, [], [(case["appgroup"], s) for s in case["created"]])
    with Engine(0) as e:
        e.load_network_objects(nodes, pods, ag, nt)
        a, b = np.divmod(np.arange(n * n), n)
        less = e.toposort_less(pods, a, b).reshape(n, n)
    assert less.diagonal().all()                           # Less is orderP1 <= orderP2 (topologicalsort.go:131)
    off = less & ~np.eye(n, dtype=bool)
    assert not (off & off.T).any()                         # distinct workloads: a strict total order off the diagonal
    order = np.argsort(off.sum(axis=0), kind="stable")     # rank = number of predecessors
    assert [case["created"][i] for i in order] == case["popped"]


@pytest.mark.parametrize("kernel", ["class_table", "generic"])
@pytest.mark.parametrize("n_nodes,n_pods,seed,ppg", [(500, 300, 1, 30), (64, 40, 2, 5), (1, 3, 3, 1), (1030, 129, 4, 10), (257, 200, 5, 200)])
def test_differential(gpu_required, hdr, oracle, kernel, n_nodes, n_pods, seed, ppg):
    """both table sweeps (the class-table kernel and the per-node one it replaces) against the oracle"""
    snap = synth.network_snapshot(hdr, n_nodes, n_pods, seed=seed, pods_per_group=ppg)
    with Engine(0) as e:
        if kernel == "generic":
            e.force_reference_kernels(NETOVERHEAD)
        e.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
        assert e.kernel_path(NETOVERHEAD) == (1 if kernel == "class_table" else 0)
        e.eval(mask_of(NETOVERHEAD))
        e.sync()
        osnap = oracle.Snapshot(snap["nodes"], snap["pods"], appgroups=snap["appgroups"], nettopo=snap["nettopo"])
        want_status = osnap.filter_rows(NETOVERHEAD)
        raw, norm = osnap.score_rows(NETOVERHEAD)
        got_status = e.all_status(NETOVERHEAD)
        assert np.array_equal(got_status, want_status)
        got = e.all_scores(NETOVERHEAD).astype(np.int64)
        bad = np.argwhere(got != norm)
        assert bad.size == 0, f"{len(bad)} mismatches, first {[(int(p), int(n), int(got[p, n]), int(norm[p, n])) for p, n in bad[:5]]}"
        # raw accumulated cost on every node (also the filtered ones), and satisfied/violated counts
        import ctypes as C
        i64p = C.POINTER(C.c_int64)
        for r in sorted({0, n_pods // 2, n_pods - 1}):
            sat, vio, cost = (np.zeros(n_nodes, np.int64) for _ in range(3))
            oracle.lib().orc_net_prefilter(snap["nodes"].ref(), snap["pods"].ref(), snap["appgroups"].ref(), snap["nettopo"].ref(), r,
                                           sat.ctypes.data_as(i64p), vio.ctypes.data_as(i64p), cost.ctypes.data_as(i64p))
            assert np.array_equal(e.raw(NETOVERHEAD, r, 0), cost)
            assert np.array_equal(e.raw(NETOVERHEAD, r, 1), sat)
            assert np.array_equal(e.raw(NETOVERHEAD, r, 2), vio)
        # TopologicalSort.Less on random pairs, through the flattened keys
        rng = np.random.default_rng(seed)
        a = rng.integers(0, n_pods, 500)
        b = rng.integers(0, n_pods, 500)
        want = [bool(oracle.lib().orc_toposort_less(snap["pods"].ref(), snap["appgroups"].ref(), int(x), int(y))) for x, y in zip(a, b)]
        assert e.toposort_less(snap["pods"], a, b).tolist() == want


def test_host_missing_from_snapshot_is_an_error(gpu_required, hdr):
    """a scheduled pod whose host is not in the snapshot: PreFilter returns fwk.Error (networkoverhead.go:258)"""
    nodes, pods, ag, nt = build(hdr, [("p2", "n-unknown")], [("basic", "p1")])
    with Engine(0) as e:
        e.load_network_objects(nodes, pods, ag, nt)
        e.eval(mask_of(NETOVERHEAD))
        e.sync()
        assert (e.status(NETOVERHEAD, 0) == 255).all()


# ------------------------------------------------------------------ full size (config #4): sampled rows + properties
def test_config4_full_size_properties(gpu_required, hdr, oracle):
    n_nodes, n_pods = 10_000, 200_000
    snap = synth.network_snapshot(hdr, n_nodes, n_pods)
    with Engine(0) as e:
        e.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
        assert e.kernel_path(NETOVERHEAD) == 1
        e.eval(mask_of(NETOVERHEAD))
        e.sync()
        osnap = oracle.Snapshot(snap["nodes"], snap["pods"], appgroups=snap["appgroups"], nettopo=snap["nettopo"])
        rng = np.random.default_rng(11)
        rows = sorted(set(rng.integers(0, n_pods, 12).tolist()) | {0, n_pods - 1})
        for r in rows:
            assert np.array_equal(e.status(NETOVERHEAD, r), osnap.filter_rows(NETOVERHEAD, r, r + 1)[0]), r
            assert np.array_equal(e.scores(NETOVERHEAD, r).astype(np.int64), osnap.score_rows(NETOVERHEAD, r, r + 1)[1][0]), r
        # a row depends on the pod only through its (AppGroup, selector): equal workloads -> equal rows
        ag, sel = snap["pods"].array("appgroup"), snap["pods"].array("selector")
        key = ag.astype(np.int64) * 64 + sel
        order = np.argsort(key, kind="stable")
        same = np.flatnonzero((key[order][1:] == key[order][:-1]))[:: 4001][:24]
        assert same.size > 0
        for i in same:
            a, b = int(order[i]), int(order[i + 1])
            assert np.array_equal(e.scores(NETOVERHEAD, a), e.scores(NETOVERHEAD, b)) and np.array_equal(e.status(NETOVERHEAD, a), e.status(NETOVERHEAD, b))
        # NormalizeScore: every evaluated row spans exactly [.., 100] with 100 at its cheapest feasible node, or is all zero
        for r in rows:
            sc, st = e.scores(NETOVERHEAD, r), e.status(NETOVERHEAD, r)
            assert sc.max() in (0, 100) and (sc[st != 0] == 0).all()


# ------------------------------------------------------------------ TopologicalSort as a device sort (spx_sort_keys)
def _violations(oracle, snap_pods, ag, perm):
    import ctypes as C
    perm = np.ascontiguousarray(perm, dtype=np.int32)
    return int(oracle.lib().orc_toposort_order_violations(snap_pods.ref(), ag.ref(), perm.ctypes.data_as(C.POINTER(C.c_int32)), len(perm)))


@pytest.mark.parametrize("case", GN.QUEUE_ORDER_CASES, ids=lambda c: f"L{c['line']}")
def test_sort_queue_golden(gpu_required, hdr, oracle, case):
    """test/integration/topologicalsort_test.go:253-342: pods of one AppGroup, equal priority — the queue pops them in
    topology-index order, and distinct indexes make that order unique, so the device sort must reproduce it exactly"""
    nodes, pods, ag, nt = build(hdr, [], [(case["appgroup"], s) for s in case["created"]])
    with Engine(0) as e:
        e.load_network_objects(nodes, pods, ag, nt)
        perm = e.sort_queue(pods)
    assert [case["created"][i] for i in perm] == case["popped"]
    assert _violations(oracle, pods, ag, perm) == 0


@pytest.mark.parametrize("n_pods,ppg,seed,mode", [(1, 1, 1, "mixed"), (63, 5, 2, "mixed"), (64, 64, 3, "mixed"), (1025, 30, 4, "mixed"), (5000, 7, 5, "equal_ts"),
                                                  (4097, 300, 6, "one_priority"), (20000, 100, 7, "no_groups"), (3001, 11, 8, "negative")])
def test_sort_queue_property(gpu_required, hdr, oracle, n_pods, ppg, seed, mode):
    """every adjacent pair of the returned order satisfies the reference's Less (or is a PrioritySort tie), on queues that
    stress each key: interleaved AppGroups, timestamp ties, a single priority, no AppGroup at all, negative priorities and
    timestamps, pods whose selector is not in the topology order (index -1)"""
    snap = synth.network_snapshot(hdr, 40, n_pods, seed=seed, pods_per_group=ppg)
    pods = snap["pods"]
    rng = np.random.default_rng(seed)
    if mode == "equal_ts":
        pods.array("queue_ts")[:] = 1_700_000_000_000_000 + rng.integers(0, 3, n_pods)
    elif mode == "one_priority":
        pods.array("priority")[:] = 7
        pods.array("queue_ts")[:] = rng.permutation(n_pods).astype(np.int64) * 977
    elif mode == "no_groups":
        pods.array("appgroup")[:] = -1
    elif mode == "negative":
        pods.array("priority")[:] = rng.integers(-2**31, 2**31 - 1, n_pods, dtype=np.int64).astype(np.int32)
        pods.array("queue_ts")[:] = rng.integers(-2**62, 2**62, n_pods)
        pods.array("selector")[rng.random(n_pods) < 0.1] = 10_000  # not listed in Status.TopologyOrder -> FindPodOrder = -1
    else:
        pods.array("queue_ts")[:] = rng.permutation(n_pods).astype(np.int64) * 1000 + 1_700_000_000_000_000
    with Engine(0) as e:
        e.load_network_objects(snap["nodes"], pods, snap["appgroups"], snap["nettopo"])
        perm = e.sort_queue(pods)
        # the per-pod keys the sort used are the ones the pairwise comparator agrees with the oracle on
        a, b = rng.integers(0, n_pods, 300), rng.integers(0, n_pods, 300)
        want = [bool(oracle.lib().orc_toposort_less(pods.ref(), snap["appgroups"].ref(), int(x), int(y))) for x, y in zip(a, b)]
        assert e.toposort_less(pods, a, b).tolist() == want
    assert sorted(perm.tolist()) == list(range(n_pods))
    assert _violations(oracle, pods, snap["appgroups"], perm) == 0
    if mode == "no_groups":  # plain PrioritySort: the order is the lexicographic one
        key = np.lexsort((pods.array("queue_ts"), -pods.array("priority").astype(np.int64)))
        assert np.array_equal(perm, key.astype(np.int32))


def test_sort_queue_config4(gpu_required, hdr, oracle):
    """BASELINE config #4's queue: 200k pending pods"""
    n_pods = 200_000
    snap = synth.network_snapshot(hdr, 10_000, n_pods)
    rng = np.random.default_rng(3)
    snap["pods"].array("queue_ts")[:] = rng.permutation(n_pods).astype(np.int64) * 1000 + 1_700_000_000_000_000  # arrival order != row order
    with Engine(0) as e:
        e.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
        perm = e.sort_queue(snap["pods"])
        ms = e.last_eval_ms()
    assert _violations(oracle, snap["pods"], snap["appgroups"], perm) == 0
    assert ms < 20.0, ms
