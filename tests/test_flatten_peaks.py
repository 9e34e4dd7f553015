"""Host flatteners of Peaks (scheduler-plugins_amd/host/flatten_peaks.cc) against the oracle.  CPU only."""
import ctypes as C

import numpy as np

import scheduler_plugins_amd as spx
from golden import peaks as GP
from helpers import power_models
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd import synth


def flatten_nodes(nodes, metrics, models):
    n = nodes.struct.n_nodes
    cols = {"cap_cpu_milli": np.zeros(n, np.int64), "cpu_util": np.zeros(n, np.float64), "valid": np.zeros(n, np.uint8),
            "k1": np.zeros(n, np.float64), "k2": np.zeros(n, np.float64)}
    fn = spx.lib().spx_flatten_peaks_nodes
    assert fn(nodes.ref(), metrics.ref(), models.ref() if models else None, *[v.ctypes.data_as(t) for v, t in zip(cols.values(), fn.argtypes[3:])]) == 0
    return cols


def test_pods_match_oracle(hdr, oracle):
    pods = synth.synth_pods(hdr, 3000, seed=8)
    cpu = np.zeros(3000, np.int64)
    assert spx.lib().spx_flatten_peaks_pods(pods.ref(), cpu.ctypes.data_as(C.POINTER(C.c_int64))) == 0
    f = oracle.lib().orc_get_resource_request_quantity_cpu_milli
    f.restype = C.c_int64
    assert cpu.tolist() == [f(pods.ref(), i) for i in range(3000)]
    assert (cpu == 0).any() and (cpu > 0).any()


def test_first_avg_or_latest_metric_wins(hdr):
    res = O.Resources()
    nodes = O.build_node_objects(hdr, res, [O.node(GP.NODE)] * 5)
    metrics = O.build_metrics_objects(hdr, 5, {
        0: [("CPU", "STD", 9), ("CPU", "Latest", 11), ("CPU", "AVG", 22)],   # first of AVG/Latest: 11 (TLP would take 22)
        1: [("Memory", "AVG", 50)],                                          # no cpu metric
        2: None,                                                             # nil metrics slice
        3: [("CPU", "", 40), ("CPU", "AVG", 33)],                            # the empty operator does not count here
    })                                                                       # node 4: not in the map
    cols = flatten_nodes(nodes, metrics, power_models(hdr, [GP.POWER_MODEL, None, GP.POWER_MODEL, None, None]))
    assert cols["valid"].tolist() == [1, 0, 0, 1, 0]
    assert cols["cpu_util"].tolist() == [11, 0, 0, 33, 0]
    assert cols["cap_cpu_milli"].tolist() == [1000] * 5
    assert cols["k1"].tolist() == [GP.POWER_MODEL["k1"], 0, GP.POWER_MODEL["k1"], 0, 0]
    nilmap = O.build_metrics_objects(hdr, 5, None)
    assert not flatten_nodes(nodes, nilmap, None)["valid"].any()
