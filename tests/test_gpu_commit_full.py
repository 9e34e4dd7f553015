"""spx_commit_sequential and the full profile at the sizes BASELINE.json states, against the CPU oracle — not against a second
form of the same device code:

  * config #2's chain: ALL 100 000 decisions (node, weighted score, tie-set size) of the one-workgroup register-resident loop at
    10 000 nodes equal oracle/orc_commit.c's one-pod-at-a-time cycle (the ScheduledPodsCache grows by one entry per pod);
  * the full profile through the cooperative persistent kernel at config #5's node count: 8 192 pods x 20 000 nodes = 79
    workgroups exchanging granules across XCDs, every decision and every unschedulable verdict equal the oracle's cycle with the
    NRT assumed store, the AppGroup scheduled lists, ElasticQuota Used / nominated pods and trimaran's cache as mutable state;
  * config #5 WHOLE on one device: 500 000 pods x 20 000 nodes in one engine (seven 10 GB tables), >= 2 000 rows spread over the
    batch (first, last, and the rows either side of every 2^31-byte boundary of a table) compared cell by cell, with their decisions.

orc_commit.c itself is pinned on CPU by tests/test_oracle_commit.py (against per-pod rebuilds of every object table).
"""
import time

import numpy as np
import pytest

from helpers import ALLOCATABLE, CAPACITY, LVRB, NETOVERHEAD, NRT, TLP, lvrb_params, tlp_params
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd import synth
from scheduler_plugins_amd.engine import Engine, mask_of

pytestmark = pytest.mark.gpu


def _compare(got_node, got_score, got_ties, want):
    placed = want["node"] >= 0
    bad_node = int((got_node != want["node"]).sum())
    bad_score = int((got_score[placed] != want["score"][placed]).sum())
    bad_ties = int((got_ties != want["ties"]).sum())
    first = np.flatnonzero((got_node != want["node"]) | (got_ties != want["ties"]))
    return {"node": bad_node, "score": bad_score, "ties": bad_ties, "first_bad_pod": int(first[0]) if first.size else -1}


def test_config2_chain_every_decision(gpu_required, hdr, oracle):
    n_nodes, n_pods = 10_000, 100_000
    snap = synth.trimaran_snapshot(hdr, n_nodes, n_pods)
    mask = mask_of(ALLOCATABLE, TLP)
    with Engine(0) as e:
        e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        node, score, ties, missing = e.commit_sequential(mask)
        assert e.commit_path() == 1
        alloc_params = e.alloc_params
        missing0 = e.flatten_trimaran_nodes(snap["nodes"], snap["metrics"], snap["assigned"])["tlp_missing_milli"].copy()
        pod_cpu = e.flatten_trimaran_pods(snap["pods"])["tlp_pod_milli"].copy()
    osnap = oracle.Snapshot(snap["nodes"], snap["pods"], rc=snap["rc"], metrics=snap["metrics"], assigned=snap["assigned"], alloc_params=alloc_params,
                            tlp_params=tlp_params(hdr), lvrb_params=lvrb_params(hdr))
    t0 = time.time()
    want = oracle.commit_sequential(osnap, mask, bind_ts=int(snap["metrics"].struct.window_end) + 1)
    print(f"oracle cycle: {n_pods} pods x {n_nodes} nodes in {time.time() - t0:.1f} s on {oracle.usable_cpus()} threads")
    bad = _compare(node, score, ties, want)
    assert (bad["node"], bad["score"], bad["ties"]) == (0, 0, 0), bad
    assert (want["node"] >= 0).all() and len(set(node.tolist())) > 1000
    # the device's missing-utilisation column after the last commit = the snapshot's + the predictions of the pods the ORACLE bound
    grown = missing0.astype(np.int64)
    np.add.at(grown, want["node"], pod_cpu.astype(np.int64))
    assert np.array_equal(missing, grown)


def _full(hdr, n_nodes, n_pods):
    snap = synth.full_snapshot(hdr, n_nodes, n_pods, seed=synth.SEED, quota_sized_for_batch=True)  # bench.py's config #5 workloads
    snap["nrt_params"] = O.nrt_params(hdr, O.Resources(), "LeastAllocated")
    return snap


def _load_full(e, snap):
    e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
    e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], snap["nrt_params"])
    e.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
    e.load_quota_objects(snap["pods"], snap["rc"], snap["quota"])


ALLP = (ALLOCATABLE, TLP, LVRB, NRT, NETOVERHEAD, CAPACITY)


@pytest.mark.parametrize("weights", [None, {ALLOCATABLE: 1, TLP: 2, LVRB: 1, NRT: 3, NETOVERHEAD: 2}])
def test_full_profile_coop_8192_pods_20000_nodes(gpu_required, hdr, oracle, weights):
    n_nodes, n_pods = 20_000, 8_192
    snap = _full(hdr, n_nodes, n_pods)
    mask = mask_of(*ALLP)
    with Engine(0) as e:
        _load_full(e, snap)
        if weights:
            e.set_plugin_weights(weights)
        node, score, ties, _ = e.commit_sequential(mask)
        assert e.commit_path() == 3   # the cooperative persistent kernel: 79 workgroups
        alloc_params = e.alloc_params
    osnap = oracle.Snapshot(snap["nodes"], snap["pods"], rc=snap["rc"], metrics=snap["metrics"], assigned=snap["assigned"], alloc_params=alloc_params,
                            tlp_params=tlp_params(hdr), lvrb_params=lvrb_params(hdr), nrt=snap["nrt"], nrt_params=snap["nrt_params"],
                            appgroups=snap["appgroups"], nettopo=snap["nettopo"])
    t0 = time.time()
    want = oracle.commit_sequential(osnap, mask, weights, quota=snap["quota"], bind_ts=int(snap["metrics"].struct.window_end) + 1)
    print(f"oracle cycle: {n_pods} pods x {n_nodes} nodes, full profile, in {time.time() - t0:.1f} s on {oracle.usable_cpus()} threads; "
          f"{int((want['node'] < 0).sum())} unschedulable ({int((want['verdict'] == 255).sum())} without a feasible node)")
    bad = _compare(node, score, ties, want)
    assert (bad["node"], bad["score"], bad["ties"]) == (0, 0, 0), bad
    n_unsched = int((want["node"] < 0).sum())
    assert 0 < n_unsched < n_pods // 4 and len(set(node.tolist())) > 2000


def test_config5_whole_on_one_device_sampled_rows(gpu_required, hdr, oracle):
    """BASELINE config #5 as stated — 20 000 nodes x 500 000 pods, the full profile — in ONE engine on one MI355X (row J1 of the
    round-4 review): the N = 1 anchor of the strong-scaling curve.  Rows compared with the oracle, every cell of every table plus the
    decision: 2 048 rows evenly spread (incl. the first and the last), and the rows around every multiple of 2^31 bytes in a table
    (where a 32-bit offset would wrap)."""
    n_nodes, n_pods = 20_000, 500_000
    snap = _full(hdr, n_nodes, n_pods)
    weights = {ALLOCATABLE: 1, TLP: 2, LVRB: 1, NRT: 3, NETOVERHEAD: 2}
    threads = oracle.usable_cpus()
    with Engine(0) as e:
        _load_full(e, snap)
        e.set_plugin_weights(weights)
        e.eval(mask_of(*ALLP))
        e.eval_best(mask_of(*ALLP))
        e.sync()
        stride = e.score_table(TLP)[1]
        rows = set(np.linspace(0, n_pods - 1, 2048).astype(np.int64).tolist())
        k = 1
        while k * (1 << 31) < n_pods * stride:
            r = (k * (1 << 31)) // stride
            rows.update(x for x in (r - 1, r, r + 1) if 0 <= x < n_pods)
            k += 1
        rows = np.array(sorted(rows), dtype=np.int64)
        assert rows[0] == 0 and rows[-1] == n_pods - 1 and rows.size >= 2048 + 3 * 4
        got = {p: np.stack([e.all_scores(p, int(r), int(r) + 1)[0] for r in rows]) for p in weights}
        got_st = {p: np.stack([e.all_status(p, int(r), int(r) + 1)[0] for r in rows]) for p in (NRT, NETOVERHEAD)}
        node, score, ties, feas = (a[rows] for a in e.best())
        pre_got = e.prefilter(CAPACITY)[rows]
        alloc_params = e.alloc_params
    sub = synth.take_pods(hdr, snap["pods"], rows)   # the sampled pods as a table of their own: the oracle walks them contiguously
    osnap = oracle.Snapshot(snap["nodes"], sub, rc=snap["rc"], metrics=snap["metrics"], assigned=snap["assigned"], alloc_params=alloc_params,
                            tlp_params=tlp_params(hdr), lvrb_params=lvrb_params(hdr), nrt=snap["nrt"], nrt_params=snap["nrt_params"],
                            appgroups=snap["appgroups"], nettopo=snap["nettopo"])
    pre = np.array([oracle.lib().orc_capacity_prefilter(snap["pods"].ref(), snap["rc"].ref(), snap["quota"].ref(), int(r)) for r in rows], dtype=np.uint8)
    assert np.array_equal(pre_got, pre)
    R = rows.size
    nrt_st = osnap.filter_rows(NRT, 0, R, threads=threads)
    net_st = osnap.filter_rows(NETOVERHEAD, 0, R, threads=threads)
    bad = {"nrt_status": int((got_st[NRT] != nrt_st).sum()), "net_status": int((got_st[NETOVERHEAD] != net_st).sum())}
    want = {p: osnap.score_rows(p, 0, R, threads=threads, want_norm=False)[0].clip(0, 255) for p in (TLP, LVRB, NRT)}
    feasible = (nrt_st == 0) & (net_st == 0)
    want[NETOVERHEAD] = osnap.score_rows(NETOVERHEAD, 0, R, mask=(nrt_st == 0).astype(np.uint8), threads=threads, want_raw=False)[1]
    want[ALLOCATABLE] = osnap.score_rows(ALLOCATABLE, 0, R, mask=feasible.astype(np.uint8), threads=threads, want_raw=False)[1]
    for p in weights:
        bad[p] = int((got[p].astype(np.int64) != want[p]).sum())
    total = sum(weights[p] * want[p] for p in weights)
    total[~feasible] = -1
    best = total.max(axis=1)
    none = (pre != 0) | (best < 0)
    ok = np.where(none, (node == -1) & (ties == 0),
                  (node == total.argmax(axis=1)) & (score == best) & (ties == (total == best[:, None]).sum(axis=1)) & (feas == feasible.sum(axis=1)))
    bad["best"] = int((~ok).sum())
    assert not any(bad.values()), bad
    assert 0 < none.sum() < R and 0 < (nrt_st != 0).sum() and 0 < (net_st != 0).sum()
