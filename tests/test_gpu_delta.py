"""Snapshot deltas (spx_update_trimaran_nodes / spx_update_nrt_nodes, SURVEY 8d "upload deltas"): replacing the rows of the
changed nodes in place must leave exactly the tables a full re-upload of the new snapshot leaves."""
import numpy as np
import pytest

from helpers import ALLOCATABLE, LVRB, NRT, TLP
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd import synth
from scheduler_plugins_amd.engine import Engine, mask_of

pytestmark = pytest.mark.gpu


def test_trimaran_node_delta_equals_full_upload(gpu_required, hdr):
    n_nodes, n_pods = 3000, 700
    old = synth.trimaran_snapshot(hdr, n_nodes, n_pods, seed=11)
    new = synth.trimaran_snapshot(hdr, n_nodes, n_pods, seed=12)  # other metrics, other bind-time cache
    mask = mask_of(ALLOCATABLE, TLP, LVRB)
    with Engine(0) as ref, Engine(0) as e:
        cols_old = e.flatten_trimaran_nodes(old["nodes"], old["metrics"], old["assigned"])
        cols_new = e.flatten_trimaran_nodes(old["nodes"], new["metrics"], new["assigned"])
        rng = np.random.default_rng(1)
        idx = rng.choice(n_nodes, 37, replace=False)
        mixed = {k: v.copy() for k, v in cols_old.items()}
        for k in mixed:
            mixed[k][idx] = cols_new[k][idx]
        for eng, cols in ((ref, mixed), (e, cols_old)):
            eng.upload_alloc_nodes(eng.flatten_alloc_nodes(old["nodes"], old["rc"]))
            eng.upload_trimaran_nodes(cols)
            eng.upload_trimaran_pods(eng.flatten_trimaran_pods(old["pods"]))
        e.eval(mask)
        e.sync()
        before = e.all_scores(TLP)
        e.update_trimaran_nodes(idx, cols_new)
        with pytest.raises(Exception):
            e.all_scores(TLP)  # tables computed from the old rows are stale
        e.eval(mask)
        ref.eval(mask)
        e.sync(), ref.sync()
        for p in (ALLOCATABLE, TLP, LVRB):
            assert np.array_equal(e.all_scores(p), ref.all_scores(p))
        changed = np.flatnonzero((e.all_scores(TLP) != before).any(axis=0))
        assert len(changed) > 0 and set(changed.tolist()) <= set(idx.tolist())  # only the touched nodes' columns moved


@pytest.mark.parametrize("strategy", ["LeastAllocated", "BalancedAllocation", "LeastNUMANodes"])
@pytest.mark.parametrize("kernel", ["fast", "reference"])
def test_nrt_node_delta_equals_full_upload(gpu_required, hdr, strategy, kernel):
    n_nodes, n_pods = 700, 300
    old = synth.nrt_snapshot(hdr, n_nodes, n_pods, seed=31)
    new = synth.nrt_snapshot(hdr, n_nodes, n_pods, seed=32)  # same node sizes (seeded separately), other zones / policies / costs
    params = O.nrt_params(hdr, O.Resources(), strategy)
    with Engine(0) as ref, Engine(0) as e:
        f_old = e.flatten_nrt(old["nodes"], old["nrt"], old["rc"], old["pods"], params)
        f_new = e.flatten_nrt(old["nodes"], new["nrt"], old["rc"], old["pods"], params)
        rng = np.random.default_rng(2)
        idx = rng.choice(n_nodes, 23, replace=False)
        per = {"flags": 1, "max_numa": 1, "n_zones": 1, "zone_id": 8, "zone_present": 8, "zone_avail": 8 * f_old["R"], "zone_cost": 64,
               "min_avg_dist": 8, "node_present": 1}
        f_mix = dict(f_old)
        f_mix["nodes"] = {k: v.copy() for k, v in f_old["nodes"].items()}
        for k, w in per.items():
            f_mix["nodes"][k].reshape(n_nodes, w)[idx] = f_new["nodes"][k].reshape(n_nodes, w)[idx]
        ref.upload_nrt(f_mix)
        e.upload_nrt(f_old)
        if kernel == "reference":
            ref.force_reference_kernels(NRT), e.force_reference_kernels(NRT)
        e.eval(mask_of(NRT))
        e.sync()
        e.update_nrt_nodes(idx, f_new)
        e.eval(mask_of(NRT))
        ref.eval(mask_of(NRT))
        e.sync(), ref.sync()
        assert e.kernel_path(NRT) == ref.kernel_path(NRT) == (1 if kernel == "fast" else 0)
        assert np.array_equal(e.all_status(NRT), ref.all_status(NRT))
        assert np.array_equal(e.all_scores(NRT), ref.all_scores(NRT))
        for r in (0, n_pods - 1):
            assert np.array_equal(e.raw(NRT, r), ref.raw(NRT, r))


def test_delta_refuses_a_node_listed_twice_and_invalidates_dependent_tables(gpu_required, hdr):
    """a duplicated index would scatter one node's row twice in no particular order: both deltas refuse it and leave the engine as
    it was; an accepted NRT delta marks every table stale (Allocatable normalises over the feasible nodes NRT's status names)"""
    n_nodes, n_pods = 300, 64
    snap = synth.nrt_snapshot(hdr, n_nodes, n_pods, seed=41)
    tri = synth.trimaran_snapshot(hdr, n_nodes, n_pods, seed=41)
    params = O.nrt_params(hdr, O.Resources(), "LeastAllocated")
    with Engine(0) as e:
        f = e.flatten_nrt(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], params)
        e.upload_nrt(f)
        cols = e.flatten_trimaran_nodes(tri["nodes"], tri["metrics"], tri["assigned"])
        e.upload_alloc_nodes(e.flatten_alloc_nodes(tri["nodes"], tri["rc"]))
        e.upload_trimaran_nodes(cols)
        e.upload_trimaran_pods(e.flatten_trimaran_pods(tri["pods"]))
        e.eval(mask_of(NRT))
        e.sync()
        before = e.all_scores(NRT)
        dup = np.array([5, 9, 5], dtype=np.int64)
        with pytest.raises(Exception, match="listed twice"):
            e.update_nrt_nodes(dup, f)
        with pytest.raises(Exception, match="listed twice"):
            e.update_trimaran_nodes(dup, cols)
        assert np.array_equal(e.all_scores(NRT), before)  # refused before anything moved: the tables still stand
        e.update_nrt_nodes(np.array([5, 9], dtype=np.int64), f)
        with pytest.raises(Exception):
            e.all_scores(NRT)
        e.eval(mask_of(NRT))
        e.sync()
        assert np.array_equal(e.all_scores(NRT), before)  # the same rows again: the same tables


def _grown_appgroups(hdr, ag, group, selector, node):
    """the AppGroup table with the pods (group[j], selector[j], node[j]) appended to their groups' placed lists"""
    from scheduler_plugins_amd._abi import Table
    G = ag.struct.n_groups
    ptr = ag.array("placed_ptr")[:G + 1].astype(np.int64)
    sel, nd = ag.array("placed_selector"), ag.array("placed_node")
    new_sel, new_nd, new_ptr = [], [], [0]
    for g in range(G):
        extra = [j for j in range(len(group)) if group[j] == g]
        new_sel += list(sel[ptr[g]:ptr[g + 1]]) + [selector[j] for j in extra]
        new_nd += list(nd[ptr[g]:ptr[g + 1]]) + [node[j] for j in extra]
        new_ptr.append(len(new_sel))
    keep = {f: ag.array(f) for f in ("wl_ptr", "wl_selector", "dep_ptr", "dep_selector", "dep_max_cost", "topo_ptr", "topo_selector", "topo_index")}
    return Table(hdr, "spx_appgroup_objects", n_groups=G, placed_ptr=np.array(new_ptr, np.int32), placed_selector=np.array(new_sel, np.int32),
                 placed_node=np.array(new_nd, np.int32), **keep)


def test_net_placed_delta_equals_full_upload(gpu_required, hdr):
    """AppGroup scheduled lists that grew between two cycles: spx_flatten_net_placed + spx_update_net_placed leave the tables a
    re-flatten + re-upload of the grown AppGroups leaves (scores, statuses, and the decisions)"""
    from helpers import NETOVERHEAD
    n_nodes, n_pods = 900, 600
    snap = synth.network_snapshot(hdr, n_nodes, n_pods, seed=21, pods_per_group=40)
    ag = snap["appgroups"]
    rng = np.random.default_rng(5)
    G = ag.struct.n_groups
    m = 57
    group = rng.integers(0, G, m).astype(np.int32)
    group[:3] = -1                                  # not AppGroup members: no entries
    wl_ptr, wl_sel = ag.array("wl_ptr"), ag.array("wl_selector")
    selector = np.array([wl_sel[rng.integers(wl_ptr[g], wl_ptr[g + 1])] if g >= 0 else 0 for g in group], np.int32)
    node = rng.integers(0, n_nodes, m).astype(np.int32)
    node[5] = -1                                    # a host outside the snapshot: the key's PreFilter turns to Error
    grown = _grown_appgroups(hdr, ag, group, selector, node)
    mask = mask_of(NETOVERHEAD)
    with Engine(0) as ref, Engine(0) as e:
        ref.load_network_objects(snap["nodes"], snap["pods"], grown, snap["nettopo"])
        e.load_network_objects(snap["nodes"], snap["pods"], ag, snap["nettopo"])
        e.eval(mask)
        e.sync()
        before = e.all_scores(NETOVERHEAD)
        ent = e.flatten_net_placed(snap["pods"], ag, group, selector, node)
        assert len(ent["key"]) > m // 2 and (ent["max_cost"] >= 0).any() and (ent["max_cost"] == -1).any()
        e.update_net_placed(ent)
        with pytest.raises(Exception):
            e.all_scores(NETOVERHEAD)  # computed from the old lists: stale
        e.eval(mask), ref.eval(mask)
        e.sync(), ref.sync()
        assert np.array_equal(e.all_scores(NETOVERHEAD), ref.all_scores(NETOVERHEAD))
        assert np.array_equal(e.all_status(NETOVERHEAD), ref.all_status(NETOVERHEAD))
        assert not np.array_equal(before, e.all_scores(NETOVERHEAD))
        for which in (0, 1, 2):
            assert np.array_equal(e.raw(NETOVERHEAD, 17, which), ref.raw(NETOVERHEAD, 17, which))
        e.update_net_placed(e.flatten_net_placed(snap["pods"], grown, group[10:20], selector[10:20], node[10:20]))  # a second delta on top
        ref2 = _grown_appgroups(hdr, grown, group[10:20], selector[10:20], node[10:20])
        with Engine(0) as r2:
            r2.load_network_objects(snap["nodes"], snap["pods"], ref2, snap["nettopo"])
            r2.eval(mask), e.eval(mask)
            r2.sync(), e.sync()
            assert np.array_equal(e.all_scores(NETOVERHEAD), r2.all_scores(NETOVERHEAD))
            assert np.array_equal(e.all_status(NETOVERHEAD), r2.all_status(NETOVERHEAD))


def test_quota_used_delta_equals_full_upload(gpu_required, hdr):
    """ElasticQuota Used rows replaced in place (AddPod / DeletePod between cycles) = a full upload of the new quota table"""
    from helpers import CAPACITY
    from scheduler_plugins_amd._abi import Table
    snap = synth.full_snapshot(hdr, 300, 900, seed=9, quota_sized_for_batch=True)
    quota = snap["quota"]
    NS = quota.struct.n_namespaces
    used, usedp = quota.array("used").reshape(-1, 8).copy(), quota.array("used_present").copy()
    rng = np.random.default_rng(2)
    ns = np.sort(rng.choice(NS, 9, replace=False)).astype(np.int32)
    mx = quota.array("max").reshape(-1, 8)
    used[ns] = np.where(rng.random((9, 8)) < 0.5, mx[ns], used[ns] // 2)  # some namespaces now sit at Max, others freed
    usedp[ns[:2]] = 1
    fields = {f: quota.array(f) for f in quota._keep if f not in ("used", "used_present")}
    scalars = {f: getattr(quota.struct, f) for f, _ in quota.struct._fields_ if f not in hdr.field_np["spx_quota_objects"]}
    new_quota = Table(hdr, "spx_quota_objects", used=used.reshape(-1), used_present=usedp, **fields, **scalars)
    mask = mask_of(CAPACITY)
    with Engine(0) as ref, Engine(0) as e:
        f_new = ref.flatten_quota(snap["pods"], snap["rc"], new_quota)
        ref.upload_quota(f_new)
        e.upload_quota(e.flatten_quota(snap["pods"], snap["rc"], quota))
        e.eval(mask)
        e.sync()
        before = e.prefilter(CAPACITY)
        e.update_quota_used(ns, used[ns], usedp[ns], f_new["cols"]["agg_used"], f_new["cols"]["agg_used_present"])
        with pytest.raises(Exception):
            e.prefilter(CAPACITY)
        e.eval(mask), ref.eval(mask)
        e.sync(), ref.sync()
        after = e.prefilter(CAPACITY)
        assert np.array_equal(after, ref.prefilter(CAPACITY))
        assert not np.array_equal(before, after)


def test_tlp_ambiguity_table_follows_deltas_and_params(gpu_required, hdr):
    """the TLP sweep's ambiguity table (k_tlp_amb_build) is kept across launches and rebuilt when a column it was built from, or
    the target utilisation, changes: on tie-heavy snapshots (every node carries integer-valued metrics, so exact rounding ties are
    everywhere and a stale table would leave float32-rounded bytes in them) an engine that evaluated, took a node delta, evaluated,
    changed the target and evaluated again must hold what a fresh engine with the per-cell bookkeeping (option off) computes"""
    n_nodes, n_pods = 2500, 1200
    old = synth.trimaran_snapshot(hdr, n_nodes, n_pods, seed=41, round_frac=1.0)
    new = synth.trimaran_snapshot(hdr, n_nodes, n_pods, seed=42, round_frac=1.0)
    mask = mask_of(ALLOCATABLE, TLP, LVRB)
    with Engine(0) as e:
        cols_old = e.flatten_trimaran_nodes(old["nodes"], old["metrics"], old["assigned"])
        cols_new = e.flatten_trimaran_nodes(old["nodes"], new["metrics"], new["assigned"])
        idx = np.random.default_rng(3).choice(n_nodes, 400, replace=False)
        mixed = {k: v.copy() for k, v in cols_old.items()}
        for k in mixed:
            mixed[k][idx] = cols_new[k][idx]

        both = lambda eng: np.stack([eng.all_scores(TLP), eng.all_scores(LVRB)])

        def fresh(cols, target, lv=(1.0, 1.0)):
            with Engine(0) as r:
                r.set_option("TLP_AMB_TABLE", 0)
                r.set_tlp(target_utilization=target)
                r.set_lvrb(*lv)
                r.upload_alloc_nodes(r.flatten_alloc_nodes(old["nodes"], old["rc"]))
                r.upload_trimaran_nodes(cols)
                r.upload_trimaran_pods(r.flatten_trimaran_pods(old["pods"]))
                r.stats(reset=True)
                r.eval(mask)
                r.sync()
                assert r.stats()[TLP] > 1000 and r.stats()[LVRB] > 1000   # ties all over the tables
                return np.stack([r.all_scores(TLP), r.all_scores(LVRB)])

        e.upload_alloc_nodes(e.flatten_alloc_nodes(old["nodes"], old["rc"]))
        e.upload_trimaran_nodes(cols_old)
        e.upload_trimaran_pods(e.flatten_trimaran_pods(old["pods"]))
        for _ in range(2):   # the second launch reuses the table
            e.eval(mask)
            e.sync()
            assert np.array_equal(both(e), fresh(cols_old, 40))
        e.update_trimaran_nodes(idx, cols_new)
        e.eval(mask)
        e.sync()
        assert np.array_equal(both(e), fresh(mixed, 40))
        e.set_tlp(target_utilization=57)
        e.eval(mask)
        e.sync()
        assert np.array_equal(both(e), fresh(mixed, 57))
        e.set_lvrb(0.7, 2.0)   # another sigma: LVRB's table and constants
        e.eval(mask)
        e.sync()
        assert np.array_equal(both(e), fresh(mixed, 57, (0.7, 2.0)))
        e.upload_trimaran_nodes(cols_new)   # a full re-upload
        e.eval(mask)
        e.sync()
        assert np.array_equal(both(e), fresh(cols_new, 57, (0.7, 2.0)))
