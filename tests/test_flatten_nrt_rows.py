"""spx_flatten_nrt_node_rows: the SoA rows of a subset of the nodes are the rows the full flattener writes for them (the
input of spx_update_nrt_nodes — a snapshot delta flattens only the changed NodeResourceTopology objects)"""
import numpy as np

from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd import synth
from scheduler_plugins_amd.engine import Engine


class HostOnly(Engine):
    """the flatteners are host functions of libspx.so: no device needed"""
    def __init__(self):
        import scheduler_plugins_amd as spx
        self._lib, self._hdr, self._h = spx.lib(), spx.header(), None
        self._owned = False

    def _ck(self, rc):
        assert rc == 0, rc


def test_node_rows_equal_the_full_flatten(hdr):
    snap = synth.nrt_snapshot(hdr, 900, 50, seed=4)
    e = HostOnly()
    f = e.flatten_nrt(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], O.nrt_params(hdr, O.Resources(), "LeastAllocated"))
    idx = np.random.default_rng(3).choice(900, 77, replace=False)
    rows = e.flatten_nrt_node_rows(snap["nodes"], snap["nrt"], f["slots"], idx)
    per = {"flags": 1, "max_numa": 1, "n_zones": 1, "zone_id": 8, "zone_present": 8, "zone_avail": 8 * f["R"], "zone_cost": 64,
           "min_avg_dist": 8, "node_present": 1}
    for k, w in per.items():
        assert np.array_equal(rows[k].reshape(len(idx), w), f["nodes"][k].reshape(900, w)[idx]), k
