"""spx_load_* (object tables -> SoA -> device inside the library, the calls the cgo shim makes) leave the engine in the state the
ctypes binding's own flatten + upload sequences leave it in: every table of the full profile, the decisions and the one-pod-at-a-time
commit loop agree."""
import numpy as np
import pytest

from helpers import ALLOCATABLE, CAPACITY, LVRB, NETOVERHEAD, NRT, TLP
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd import synth
from scheduler_plugins_amd.engine import Engine, mask_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("strategy", ["LeastAllocated", "BalancedAllocation"])
def test_load_c_equals_python_loaders(gpu_required, hdr, strategy):
    snap = synth.full_snapshot(hdr, 700, 400, seed=5, quota_sized_for_batch=True)
    params = O.nrt_params(hdr, O.Resources(), strategy)
    mask = mask_of(ALLOCATABLE, TLP, LVRB, NRT, NETOVERHEAD, CAPACITY)
    out = []
    for via_c in (False, True, "profile", "profile-again"):  # "profile": spx_load_profile, the four loaders side by side on library threads
        with Engine(0) as e:
            if via_c == "profile-again":  # a second snapshot load into a live engine: buffers in place, nothing left from the first
                e.load_c(snap, params, concurrent=True)
                e.eval(mask)
            if via_c:
                e.load_c(snap, params, concurrent=via_c is not True)
            else:
                e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
                e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], params)
                e.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
                e.load_quota_objects(snap["pods"], snap["rc"], snap["quota"])
            e.eval(mask)
            e.sync()
            tables = {("score", p): e.all_scores(p) for p in (ALLOCATABLE, TLP, LVRB, NRT, NETOVERHEAD)}
            tables.update({("status", p): e.all_status(p) for p in (NRT, NETOVERHEAD)})
            tables["prefilter"] = e.prefilter(CAPACITY)
            e.eval_best(mask)
            tables["best"] = np.stack([np.asarray(x, dtype=np.int64) for x in e.best()])
            node, score, ties, missing = e.commit_sequential(mask, 0, 120)
            tables["commit"] = np.stack([node.astype(np.int64), score, ties.astype(np.int64)])
            tables["missing"] = missing
            out.append(tables)
    for other in out[1:]:
        for k in out[0]:
            assert np.array_equal(out[0][k], other[k]), k
    assert (out[0]["prefilter"] != 0).any() and (out[0][("status", NRT)] != 0).any()


def test_load_trimaran_pods_equals_flatten_and_upload(gpu_required, hdr):
    """a new batch through spx_load_trimaran_pods (pinned staging, one pass) = flatten_trimaran_pods + upload_trimaran_pods"""
    snap = synth.trimaran_snapshot(hdr, 900, 500, seed=3)
    batch2 = synth.synth_pods(hdr, 500, seed=77)
    mask = mask_of(ALLOCATABLE, TLP, LVRB)
    out = []
    for fused in (False, True):
        with Engine(0) as e:
            e.load_trimaran_objects(snap["nodes"], snap.get("rc"), snap["pods"], snap["metrics"], snap.get("assigned"))
            e.eval(mask)
            if fused:
                e.load_trimaran_pods(batch2)
            else:
                e.upload_trimaran_pods(e.flatten_trimaran_pods(batch2))
            e.eval(mask)
            e.sync()
            out.append([e.all_scores(p) for p in (ALLOCATABLE, TLP, LVRB)])
    for a, b in zip(*out):
        assert np.array_equal(a, b)
