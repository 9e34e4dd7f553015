"""spx_flatten_nrt_nodes' min_avg_dist (minAvgDistanceInCombinations least_numa.go:102-138) on the synthetic 8-zone snapshot —
asymmetric costs, missing entries (255), nodes with 1/2/4 zones: the flattener grows every subset's pair sum from the subset
without its lowest zone and divides once per size; here every subset is summed from scratch, float32 like the reference."""
import itertools

import numpy as np

import scheduler_plugins_amd as spx
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd import synth
from scheduler_plugins_amd._abi import Table


def test_min_avg_distance_against_brute_force():
    hdr, lib = spx.header(), spx.lib()
    snap = synth.nrt_snapshot(hdr, 300, 8, seed=5)
    nodes, nrt = snap["nodes"], snap["nrt"]
    N = nodes.struct.n_nodes
    Z = hdr.consts["SPX_NRT_MAX_ZONES"]
    res = O.Resources()
    slot_res = np.zeros(8, np.int32)
    slot_res[:4] = [0, 1, synth.RES_HUGEPAGES_2MI, synth.RES_DEVICE]
    slots = Table(hdr, "spx_nrt_slots", n_res=4, slot_res=slot_res, slot_flags=np.zeros(8, np.uint8), slot_weight=np.ones(8, np.int64))
    outs = [np.zeros(N, np.uint8), np.zeros(N, np.int32), np.zeros(N, np.uint8), np.zeros(N * Z, np.uint8), np.zeros(N * Z, np.uint8),
            np.zeros(N * Z * 4, np.int64), np.zeros(N * Z * Z, np.int32), np.zeros(N * Z, np.float32), np.zeros(N, np.uint8)]
    fn = lib.spx_flatten_nrt_nodes
    assert fn(nodes.ref(), nrt.ref(), slots.ref(), *[o.ctypes.data_as(t) for o, t in zip(outs, fn.argtypes[3:])]) == 0
    n_zones, cost, got = outs[2], outs[6].reshape(N, Z, Z), outs[7].reshape(N, Z)
    assert set(np.unique(n_zones)) >= {0, 8} and (cost == 255).any()
    for i in range(N):
        nz = int(n_zones[i])
        for k in range(1, Z + 1):
            best = np.float32(255.0)
            for combo in itertools.combinations(range(nz), k):
                accu = int(cost[i][np.ix_(combo, combo)].sum())
                d = np.float32(accu) / np.float32(k * k)
                if d < best:
                    best = d
            assert got[i][k - 1] == best, (i, k, got[i][k - 1], best)
