"""spx_decide: the per-pod decision (best node, weighted score, ties, feasible) computed without materialising score tables
must equal spx_eval + spx_eval_best exactly — it is the same float32/float64 cell code with the argmax folded in."""
import numpy as np
import pytest

from helpers import ALLOCATABLE, CAPACITY, LROC, LVRB, NETOVERHEAD, NRT, PEAKS, TLP
from scheduler_plugins_amd import synth
from scheduler_plugins_amd import SpxError
from scheduler_plugins_amd.engine import Engine, mask_of

pytestmark = pytest.mark.gpu


def both(e, mask, rb=0, re=None):
    e.eval(mask, rb, re)
    e.eval_best(mask, rb, re)
    want = [x.copy() for x in e.best(rb, re)]
    e.decide(mask, rb, re)
    got = e.best(rb, re)
    return want, got


@pytest.mark.parametrize("n_nodes,n_pods,seed,round_frac", [(1, 1, 1, 0.0), (17, 5, 2, 1.0), (1023, 70, 3, 0.3), (1025, 130, 4, 0.0),
                                                              (3000, 257, 5, 1.0), (10_000, 300, 6, 0.1)])
@pytest.mark.parametrize("plugins,weights", [((ALLOCATABLE, TLP), {ALLOCATABLE: 1, TLP: 1}), ((TLP,), {TLP: 3}),
                                             ((ALLOCATABLE, TLP), {ALLOCATABLE: 5, TLP: 2}), ((ALLOCATABLE, TLP), {ALLOCATABLE: 0, TLP: 1})])
def test_decide_equals_eval_plus_argmax(gpu_required, hdr, n_nodes, n_pods, seed, round_frac, plugins, weights):
    snap = synth.trimaran_snapshot(hdr, n_nodes, n_pods, seed=seed, round_frac=round_frac)
    with Engine(0) as e:
        e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        e.set_plugin_weights(weights)
        want, got = both(e, mask_of(*plugins))
    for name, w, g in zip(("node", "score", "ties", "feasible"), want, got):
        assert (w == g).all(), (name, np.flatnonzero(w != g)[:5], w[w != g][:5], g[w != g][:5])
    assert (want[2] > 1).any() or round_frac == 0.0 or n_nodes < 1000   # the larger rounded snapshots do produce ties


def test_partial_rows_and_unfused_profiles(gpu_required, hdr):
    snap = synth.trimaran_snapshot(hdr, 777, 200, seed=8, round_frac=0.5)
    with Engine(0) as e:
        e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        want, got = both(e, mask_of(ALLOCATABLE, TLP), 37, 151)
        for w, g in zip(want, got):
            assert (w == g).all()
        # LVRB joins the fused form as a table that is read once (see test_decide_folds_score_only_tables)
        want, got = both(e, mask_of(ALLOCATABLE, TLP, LVRB), 11, 190)
        for w, g in zip(want, got):
            assert (w == g).all()
        # SPX_OPT_DECIDE_UNFUSED runs eval + eval_best inside: same answer by definition, tables evaluated
        e.set_option("DECIDE_UNFUSED", 1)
        want, got = both(e, mask_of(ALLOCATABLE, TLP, LVRB))
        for w, g in zip(want, got):
            assert (w == g).all()
        e.set_option("DECIDE_UNFUSED", 0)
        # a caller feasibility mask also takes the unfused route
        rng = np.random.default_rng(1)
        e.upload_feasible_mask((rng.random((200, 777)) < 0.7).astype(np.uint8))
        want, got = both(e, mask_of(ALLOCATABLE, TLP))
        for w, g in zip(want, got):
            assert (w == g).all()


@pytest.mark.parametrize("n_nodes,n_pods,seed,round_frac", [(17, 5, 2, 1.0), (1025, 130, 4, 0.0), (3000, 257, 5, 1.0), (10_000, 300, 6, 0.1)])
@pytest.mark.parametrize("plugins,weights", [((ALLOCATABLE, TLP, LVRB), {ALLOCATABLE: 1, TLP: 1, LVRB: 1}),
                                             ((TLP, LVRB), {TLP: 2, LVRB: 7}),
                                             ((ALLOCATABLE, TLP, LVRB, LROC, PEAKS), {ALLOCATABLE: 1, TLP: 1, LVRB: 1, LROC: 1, PEAKS: 1}),
                                             ((ALLOCATABLE, TLP, LROC, PEAKS), {ALLOCATABLE: 3, TLP: 1, LROC: 4, PEAKS: 2})])
def test_decide_folds_score_only_tables(gpu_required, hdr, n_nodes, n_pods, seed, round_frac, plugins, weights):
    """Profiles that add LoadVariationRiskBalancing / LowRiskOverCommitment / Peaks to {Allocatable, TargetLoadPacking}: those
    plugins' tables are evaluated and read ONCE by the fused sweep; Allocatable's and TargetLoadPacking's are never
    written and no second pass (spx_eval_best) runs.  Decisions identical to spx_eval + spx_eval_best."""
    snap = synth.trimaran_snapshot(hdr, n_nodes, n_pods, seed=seed, round_frac=round_frac, with_node_pods=True)
    snap["power_models"] = synth.synth_power_models(hdr, n_nodes, seed)

    def load(e):
        e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        if LROC in plugins:
            e.set_lroc()
            e.load_lroc_objects(snap["nodes"], snap["node_pods"], snap["pods"])
        if PEAKS in plugins:
            e.load_peaks_objects(snap["nodes"], snap["metrics"], snap["power_models"], snap["pods"])
        e.set_plugin_weights(weights)

    with Engine(0) as e:
        load(e)
        want, got = both(e, mask_of(*plugins))
    for name, w, g in zip(("node", "score", "ties", "feasible"), want, got):
        assert (w == g).all(), (name, np.flatnonzero(w != g)[:5], w[w != g][:5], g[w != g][:5])
    with Engine(0) as e:   # the fused form ran: TargetLoadPacking's table was never evaluated, the folded plugins' were
        load(e)
        e.decide(mask_of(*plugins))
        assert (e.best()[0] == want[0]).all()
        with pytest.raises(SpxError):
            e.all_scores(TLP, 0, n_pods)
        for p in plugins:
            if p not in (ALLOCATABLE, TLP):
                assert e.all_scores(p, 0, n_pods).shape == (n_pods, n_nodes)


def test_config2_size(gpu_required, hdr):
    snap = synth.trimaran_snapshot(hdr, 10_000, 100_000)
    with Engine(0) as e:
        e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        want, got = both(e, mask_of(ALLOCATABLE, TLP))
        for w, g in zip(want, got):
            assert (w == g).all()
        want, got = both(e, mask_of(ALLOCATABLE, TLP, LVRB))
    for w, g in zip(want, got):
        assert (w == g).all()


# ------------------------------------------------------------------ profiles with Filter plugins (k_decide_masked)
def _full(e, hdr, n_nodes, n_pods, seed, weights):
    from test_gpu_profile import load_all
    snap = synth.full_snapshot(hdr, n_nodes, n_pods, seed=seed, pods_per_group=20, n_namespaces=20)
    load_all(e, hdr, snap)
    e.set_plugin_weights(weights)
    return snap


@pytest.mark.parametrize("n_nodes,n_pods,seed", [(300, 200, 1), (65, 33, 2), (17, 9, 3), (2100, 150, 4)])
@pytest.mark.parametrize("weights", [{ALLOCATABLE: 1, TLP: 2, LVRB: 1, NRT: 3, NETOVERHEAD: 2}, {ALLOCATABLE: 7, TLP: 0, LVRB: 1, NRT: 1, NETOVERHEAD: 1}])
@pytest.mark.parametrize("row_workgroup", [0, 1])
def test_decide_with_filter_plugins(gpu_required, hdr, n_nodes, n_pods, seed, weights, row_workgroup):
    """the whole profile (CapacityScheduling PreFilter, NRT and NetworkOverhead Filters, five scoring plugins): Allocatable's
    feasibility-aware NormalizeScore happens inside the argmax kernel and its table is not written — same decisions"""
    from test_gpu_profile import ALL
    with Engine(0) as e:
        e.set_option("ROW_WORKGROUP", row_workgroup)  # 1: a whole workgroup per row in batch launches (what very wide rows take)
        _full(e, hdr, n_nodes, n_pods, seed, weights)
        want, got = both(e, mask_of(*ALL))
        for name, w, g in zip(("node", "score", "ties", "feasible"), want, got):
            assert (w == g).all(), (name, np.flatnonzero(w != g)[:5], w[w != g][:5], g[w != g][:5])
        assert (want[0] < 0).any() or n_pods < 30      # pods without a feasible node / rejected by PreFilter are in the batch
        with pytest.raises(SpxError):                   # nothing was written to Allocatable's table by the fused form
            e.all_scores(ALLOCATABLE)
        assert e.all_scores(NRT).shape == (n_pods, n_nodes)   # the Filter plugins' tables are there as after spx_eval
        # subsets of the profile, partial rows, a single row (a whole workgroup on the row)
        for mask in (mask_of(ALLOCATABLE, NRT), mask_of(ALLOCATABLE, TLP, NETOVERHEAD), mask_of(ALLOCATABLE, NRT, NETOVERHEAD, CAPACITY)):
            want, got = both(e, mask)
            for w, g in zip(want, got):
                assert (w == g).all()
        if n_pods > 40:
            for rb, re in ((7, 8), (11, 40), (0, n_pods)):
                want, got = both(e, mask_of(*ALL), rb, re)
                for w, g in zip(want, got):
                    assert (w == g).all()


def test_decide_with_a_caller_mask_and_the_unfused_switch(gpu_required, hdr):
    from test_gpu_profile import ALL
    weights = {ALLOCATABLE: 2, TLP: 1, LVRB: 1, NRT: 1, NETOVERHEAD: 1}
    with Engine(0) as e:
        _full(e, hdr, 500, 120, 6, weights)
        e.upload_feasible_mask((np.random.default_rng(3).random((120, 500)) < 0.6).astype(np.uint8))
        want, got = both(e, mask_of(*ALL))
        for w, g in zip(want, got):
            assert (w == g).all()
        want, got = both(e, mask_of(ALLOCATABLE, TLP))       # the caller's mask alone makes Allocatable feasibility-aware
        for w, g in zip(want, got):
            assert (w == g).all()
        e.set_option("DECIDE_UNFUSED", 1)
        want, got = both(e, mask_of(*ALL))
        for w, g in zip(want, got):
            assert (w == g).all()
        assert e.all_scores(ALLOCATABLE).shape == (120, 500)   # the unfused route is spx_eval + spx_eval_best: table written


def test_decide_with_filter_plugins_falls_back_on_a_wide_allocatable_range(gpu_required, hdr):
    """raw Allocatable scores spanning more than 2^32 have no compact form: spx_decide runs spx_eval + spx_eval_best"""
    from test_gpu_profile import ALL, load_all
    snap = synth.full_snapshot(hdr, 200, 60, seed=9, pods_per_group=20, n_namespaces=20)
    snap["nodes"].array("alloc_mem")[:] *= 64
    with Engine(0) as e:
        e.set_allocatable("Least", {1: 1})
        load_all(e, hdr, snap)
        want, got = both(e, mask_of(*ALL))
        for w, g in zip(want, got):
            assert (w == g).all()
        assert e.all_scores(ALLOCATABLE).shape == (60, 200)
