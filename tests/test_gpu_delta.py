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
