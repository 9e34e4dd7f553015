"""objects.pod_rows: a rank's slice of the pending batch as a view of the pod object table — every per-pod flattener must
write for the view exactly the rows it writes for the whole table (what MultiEngine's per-rank loading rests on)"""
import numpy as np
import pytest

from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd import synth
from helpers import tlp_params
from test_flatten_nrt_rows import HostOnly


@pytest.mark.parametrize("begin,end", [(0, 400), (137, 301), (399, 400), (0, 1)])
def test_view_flattens_like_the_slice(hdr, begin, end):
    snap = synth.full_snapshot(hdr, 120, 400, seed=9, pods_per_group=20, n_namespaces=20)
    pods = snap["pods"]
    view = O.pod_rows(hdr, pods, begin, end)
    assert view.struct.n_pods == end - begin
    e = HostOnly()
    e.tlp_params = tlp_params(hdr, 40, 1000, 1.5)
    whole, part = e.flatten_trimaran_pods(pods), e.flatten_trimaran_pods(view)
    for k in whole:
        assert np.array_equal(whole[k][begin:end], part[k]), k
    whole, part = e.flatten_lroc_pods(pods), e.flatten_lroc_pods(view)
    for k in whole:
        assert np.array_equal(whole[k][begin:end], part[k]), k
    params = O.nrt_params(hdr, O.Resources(), "LeastAllocated")
    fw = e.flatten_nrt(snap["nodes"], snap["nrt"], snap["rc"], pods, params)
    fp = e.flatten_nrt(snap["nodes"], snap["nrt"], snap["rc"], view, params)
    if fw["R"] == fp["R"] and np.array_equal(fw["slots"].array("slot_res"), fp["slots"].array("slot_res")):
        P = pods.struct.n_pods
        for k, v in fw["pods"].items():
            per = len(v) // P
            assert np.array_equal(v.reshape(P, per)[begin:end].reshape(-1), fp["pods"][k]), k
    fnw = e.flatten_network(snap["nodes"], pods, snap["appgroups"], snap["nettopo"])
    fnp = e.flatten_network(snap["nodes"], view, snap["appgroups"], snap["nettopo"])
    assert np.array_equal(fnw["cols"]["topo_order"][begin:end], fnp["cols"]["topo_order"])
