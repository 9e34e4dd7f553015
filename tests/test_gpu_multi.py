"""Multi-device layer of the C ABI (spx_multi_*): pod rows sharded across ranks inside ONE host process, node
tables replicated, all-gather of decisions and of global score / status tables.

On a one-GPU box two things can run and both do: (a) the RCCL transport with a single rank (librccl.so.1 is dlopen'ed,
ncclCommInitAll / ncclAllGather really execute), and (b) several ranks on device 0 with the peer-copy transport, which
exercises sharding, in-place binding of table slices and reassembly against the unsharded engine.  With two or more GPUs
visible the same checks run over RCCL on distinct devices."""
import ctypes as C

import numpy as np
import pytest

from helpers import ALLOCATABLE, CAPACITY, LVRB, NETOVERHEAD, NRT, TLP
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd import synth
from scheduler_plugins_amd.engine import Engine, mask_of
from scheduler_plugins_amd.multi import PEER_COPY, RCCL, MultiEngine

pytestmark = pytest.mark.gpu

ALL = (ALLOCATABLE, TLP, LVRB, NRT, NETOVERHEAD, CAPACITY)
WEIGHTS = {ALLOCATABLE: 1, TLP: 2, LVRB: 1, NRT: 3, NETOVERHEAD: 2}


def n_gpus():
    import scheduler_plugins_amd as spx
    n = C.c_int(0)
    hip = C.CDLL("libamdhip64.so")
    return n.value if hip.hipGetDeviceCount(C.byref(n)) == 0 else 0


def load_full(e, hdr, snap):
    params = O.nrt_params(hdr, O.Resources(), "LeastAllocated")
    e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
    e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], params)
    e.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
    e.load_quota_objects(snap["pods"], snap["rc"], snap["quota"])


def configs():
    out = [("rccl-1", [0], RCCL), ("copy-2-on-dev0", [0, 0], PEER_COPY), ("copy-3-on-dev0", [0, 0, 0], PEER_COPY)]
    g = n_gpus()
    if g >= 2:
        out.append((f"rccl-{min(g, 8)}", list(range(min(g, 8))), RCCL))
        out.append(("copy-2", [0, 1], PEER_COPY))
    return out


@pytest.mark.parametrize("name,devices,transport", configs(), ids=lambda v: v if isinstance(v, str) else "")
@pytest.mark.parametrize("n_nodes,n_pods", [(300, 1001), (130, 7)])
def test_full_profile_sharded_equals_unsharded(gpu_required, hdr, name, devices, transport, n_nodes, n_pods):
    snap = synth.full_snapshot(hdr, n_nodes, n_pods, seed=5, pods_per_group=20, n_namespaces=20)
    with Engine(0) as ref:
        load_full(ref, hdr, snap)
        ref.set_plugin_weights(WEIGHTS)
        ref.eval(mask_of(*ALL))
        ref.eval_best(mask_of(*ALL))
        ref.sync()
        want_best = ref.best()
        want_score = {p: ref.all_scores(p) for p in WEIGHTS}
        want_status = {p: ref.all_status(p) for p in (NRT, NETOVERHEAD)}
        want_pre = ref.prefilter(CAPACITY)
    with MultiEngine(devices, transport) as m:
        assert m.size == len(devices)
        assert m.rccl_ranks() == (len(devices) if transport == RCCL else 0)   # what RCCL itself counts (ncclCommCount)
        m.for_all(lambda e: e.set_plugin_weights(WEIGHTS))
        load_full(m, hdr, snap)
        # shards are equal contiguous ranges, the last ones possibly short or empty
        per = -(-n_pods // m.size)
        assert [m.shard(r) for r in range(m.size)] == [(min(n_pods, r * per), min(n_pods, (r + 1) * per)) for r in range(m.size)]
        for p in WEIGHTS:
            m.bind_global_table(p)
        for p in (NRT, NETOVERHEAD):
            m.bind_global_table(p, status=True)
        m.eval(mask_of(*ALL))
        m.eval_best(mask_of(*ALL))
        got_best = m.gather_best()
        for a, b in zip(got_best, want_best):
            assert np.array_equal(a, b)
        for p in WEIGHTS:
            m.allgather_table(p)
        for p in (NRT, NETOVERHEAD):
            m.allgather_table(p, status=True)
        m.sync()
        for rank in range(m.size):  # every rank holds the whole table
            for p in WEIGHTS:
                assert np.array_equal(m.global_rows(p, rank), want_score[p]), (p, rank)
            for p in (NRT, NETOVERHEAD):
                assert np.array_equal(m.global_rows(p, rank, status=True), want_status[p]), (p, rank)
        # CapacityScheduling.PreFilter: the nominated-pod self-exclusion is by batch row, so shards rebase it
        pre = np.concatenate([e.prefilter(CAPACITY) for e in m.engines if e.n_pods > 0])
        assert np.array_equal(pre, want_pre)
        ev, ga = m.last_ms()
        assert ev > 0 and ga > 0


@pytest.mark.parametrize("name,devices,transport", configs(), ids=lambda v: v if isinstance(v, str) else "")
def test_decide_sharded(gpu_required, hdr, name, devices, transport):
    """the table-less decision sweep per rank + all-gather of 20 B per pod"""
    snap = synth.trimaran_snapshot(hdr, 2100, 5003, seed=8, round_frac=0.2)
    mask = mask_of(ALLOCATABLE, TLP)
    with Engine(0) as ref:
        ref.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        ref.decide(mask)
        want = ref.best()
    with MultiEngine(devices, transport) as m:
        m.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        m.decide(mask)
        for a, b in zip(m.gather_best(), want):
            assert np.array_equal(a, b)


def test_errors(gpu_required, hdr):
    import scheduler_plugins_amd as spx
    with pytest.raises(spx.SpxError, match="distinct devices"):
        MultiEngine([0, 0], RCCL)
    with pytest.raises(spx.SpxError):
        MultiEngine([], RCCL)
    with pytest.raises(spx.SpxError, match="device_id out of range"):
        MultiEngine([0, 99], PEER_COPY)
    snap = synth.trimaran_snapshot(hdr, 100, 50, seed=1)
    with MultiEngine([0, 0], PEER_COPY) as m:
        m.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        with pytest.raises(spx.SpxError, match="no decisions"):
            m.gather_best()
        with pytest.raises(spx.SpxError, match="no global table"):
            m.allgather_table(TLP)
        m.bind_global_table(TLP)
        with pytest.raises(spx.SpxError, match="not written by the last evaluation"):
            m.allgather_table(TLP)  # bound, never evaluated: a gather would publish whatever the slab held
        m.decide(mask_of(ALLOCATABLE, TLP))  # decisions without tables
        with pytest.raises(spx.SpxError, match="not written by the last evaluation"):
            m.allgather_table(TLP)
        m.eval(mask_of(ALLOCATABLE, TLP))
        m.allgather_table(TLP)
        with pytest.raises(spx.SpxError, match="rank 0"):
            m.eval(mask_of(NRT))  # NRT tables were never uploaded: the failing rank's message comes back


# ------------------------------------------------------------------ round 5: eight ranks at the sizes BASELINE states for 8 GPUs
def _eight():
    g = n_gpus()
    return (list(range(8)), RCCL, "rccl-8") if g >= 8 else ([0] * 8, PEER_COPY, "copy-8-on-dev0")


def test_config4_eight_ranks_table_gather(gpu_required, hdr):
    """BASELINE config #4 as stated for 8 GPUs — NetworkOverhead, 10 000 nodes x 200 000 pods, 25 000 rows per rank — through spx_multi:
    the all-gathered decisions and EVERY byte of the all-gathered 2 GB score and status tables (as rank 0 and rank 7 hold them) equal
    one engine evaluating the whole batch.  Eight RCCL ranks when the box has eight devices, eight ranks on device 0 over peer copies
    otherwise (the code path, not a scaling number)."""
    devices, transport, _ = _eight()
    n_nodes, n_pods = 10_000, 200_000
    snap = synth.network_snapshot(hdr, n_nodes, n_pods, seed=synth.SEED)
    mask = mask_of(NETOVERHEAD)
    with Engine(0) as ref:
        ref.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
        ref.eval(mask)
        ref.eval_best(mask)
        ref.sync()
        want_best = ref.best()
        with MultiEngine(devices, transport) as m:
            m.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
            assert [m.shard(r) for r in range(8)] == [(r * 25_000, (r + 1) * 25_000) for r in range(8)]
            m.bind_global_table(NETOVERHEAD)
            m.bind_global_table(NETOVERHEAD, status=True)
            m.eval(mask)
            m.eval_best(mask)
            for a, b in zip(m.gather_best(), want_best):
                assert np.array_equal(a, b)
            m.allgather_table(NETOVERHEAD)
            m.allgather_table(NETOVERHEAD, status=True)
            m.sync()
            bad = 0
            for r0 in range(0, n_pods, 20_000):
                r1 = min(n_pods, r0 + 20_000)
                want_sc, want_st = ref.all_scores(NETOVERHEAD, r0, r1), ref.all_status(NETOVERHEAD, r0, r1)
                for rank in (0, 7):
                    bad += int((m.global_rows(NETOVERHEAD, rank, r0, r1) != want_sc).sum())
                    bad += int((m.global_rows(NETOVERHEAD, rank, r0, r1, status=True) != want_st).sum())
            assert bad == 0
            assert (want_best[3] < n_nodes).any()   # some rows lost nodes to the Filter


def test_config4_slab_allgather_through_rccl_one_rank(gpu_required, hdr):
    """The RCCL transport's group call at config #4's per-rank slab — 25 000 rows x 10 112 bytes = 253 MB, score and status — with the
    one rank a one-device box allows: ncclCommInitAll + the grouped in-place ncclAllGather execute at the size the eight-GPU run
    launches them with (round-5 review: one-rank RCCL had only run at toy sizes), and the table they leave is the evaluated one."""
    n_nodes, n_pods = 10_000, 25_000
    snap = synth.network_snapshot(hdr, n_nodes, n_pods, seed=synth.SEED)
    mask = mask_of(NETOVERHEAD)
    with Engine(0) as ref:
        ref.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
        ref.eval(mask)
        ref.sync()
        with MultiEngine([0], RCCL) as m:
            assert m.rccl_ranks() == 1
            m.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
            m.bind_global_table(NETOVERHEAD)
            m.bind_global_table(NETOVERHEAD, status=True)
            m.eval(mask)
            m.allgather_table(NETOVERHEAD)
            m.allgather_table(NETOVERHEAD, status=True)
            m.sync()
            assert m.last_ms()[1] >= 0
            bad = 0
            for r0 in range(0, n_pods, 5_000):
                bad += int((m.global_rows(NETOVERHEAD, 0, r0, r0 + 5_000) != ref.all_scores(NETOVERHEAD, r0, r0 + 5_000)).sum())
                bad += int((m.global_rows(NETOVERHEAD, 0, r0, r0 + 5_000, status=True) != ref.all_status(NETOVERHEAD, r0, r0 + 5_000)).sum())
            assert bad == 0


def test_config5_eight_ranks_decisions(gpu_required, hdr):
    """BASELINE config #5 as stated for 8 GPUs — the full profile, 20 000 nodes x 500 000 pods, 62 500 rows per rank (seven 1.25 GB
    tables each) — through spx_multi with the practical exchange (per-pod decisions, 20 B per pod): all 500 000 decisions, and the
    CapacityScheduling verdicts with the nominated pods' batch rows rebased per shard, equal ONE engine holding the whole batch (the
    N = 1 anchor of tests/test_gpu_commit_full.py).  ~150 GB of tables on one device when the box has no eight."""
    devices, transport, _ = _eight()
    n_nodes, n_pods = 20_000, 500_000
    snap = synth.full_snapshot(hdr, n_nodes, n_pods, seed=synth.SEED, quota_sized_for_batch=True)
    with Engine(0) as ref:
        load_full(ref, hdr, snap)
        ref.set_plugin_weights(WEIGHTS)
        ref.decide(mask_of(*ALL))
        ref.sync()
        want_best = ref.best()
        want_pre = ref.prefilter(CAPACITY)
    with MultiEngine(devices, transport) as m:
        m.for_all(lambda e: e.set_plugin_weights(WEIGHTS))
        load_full(m, hdr, snap)
        assert [m.shard(r) for r in range(8)] == [(r * 62_500, (r + 1) * 62_500) for r in range(8)]
        m.decide(mask_of(*ALL))
        got = m.gather_best()
        for a, b, what in zip(got, want_best, ("node", "score", "ties", "feasible")):
            assert np.array_equal(a, b), (what, int((a != b).sum()))
        pre = np.concatenate([e.prefilter(CAPACITY) for e in m.engines])
        assert np.array_equal(pre, want_pre)
        assert 0 < (want_best[0] < 0).sum() < n_pods
