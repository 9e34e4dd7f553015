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


def test_zone_costs_by_numa_id_including_a_repeated_id(hdr):
    """the distance matrix is indexed by list position; a Costs entry names a NUMA id (least_numa.go:127-132).  Unique ids take the
    one-walk fill, an id carried by two zones the column-by-column walk in which BOTH zones take the entry; 255 where no entry exists"""
    res = O.Resources()
    costs = {"node-0": 10, "node-1": 20, "node-5": 31}
    def zone(name):
        return {"name": name, "type": "Node", "resources": {"cpu": "4", "memory": "8Gi"}, "costs": dict(costs)}
    cases = {
        "unique": (["node-1", "node-0", "node-5"], [[20, 10, 31]] * 3),
        "repeated": (["node-0", "node-1", "node-1"], [[10, 20, 20]] * 3),
        "no entry": (["node-0", "node-2"], [[10, 255]] * 2),
    }
    for name, (zones, want) in cases.items():
        nrts = O.build_nrt_objects(hdr, res, [O.nrt([zone(z) for z in zones], ["SingleNUMANodeContainerLevel"])])
        nodes = O.build_node_objects(hdr, res, [O.node({"cpu": "16", "memory": "32Gi"}, {"cpu": "16", "memory": "32Gi"})])
        pods = O.build_pod_objects(hdr, res, [O.pod([O.container({"cpu": "1"})])])
        f = HostOnly().flatten_nrt(nodes, nrts, res.table(hdr), pods, O.nrt_params(hdr, res, "LeastNUMANodes"))
        got = f["nodes"]["zone_cost"].reshape(8, 8)
        nz = len(zones)
        assert got[:nz, :nz].tolist() == want, (name, got[:nz, :nz])
        outside = got.copy()
        outside[:nz, :nz] = 255
        assert (outside == 255).all(), name
