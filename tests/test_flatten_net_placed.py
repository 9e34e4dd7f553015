"""spx_flatten_net_placed (host, no GPU): the entries a batch of newly placed AppGroup pods adds to the workload keys' pair lists
equal the difference between spx_flatten_net_keys of the grown AppGroup table and of the old one."""
import ctypes as C

import numpy as np

import scheduler_plugins_amd as spx
from scheduler_plugins_amd import synth
from scheduler_plugins_amd._abi import Table


def _keys(lib, pods, ag):
    i32p, i64p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
    nk, npairs = C.c_int32(), C.c_int64()
    assert lib.spx_flatten_net_keys(pods.ref(), ag.ref(), C.byref(nk), C.byref(npairs), None, None, None, None, None, None) == 0
    P = pods.struct.n_pods
    c = dict(pod_key=np.zeros(P, np.int32), topo=np.zeros(P, np.int32), flag=np.zeros(nk.value, np.uint8), ptr=np.zeros(nk.value + 1, np.int32),
             node=np.zeros(max(npairs.value, 1), np.int32), cost=np.zeros(max(npairs.value, 1), np.int64))
    assert lib.spx_flatten_net_keys(pods.ref(), ag.ref(), C.byref(nk), C.byref(npairs), c["pod_key"].ctypes.data_as(i32p), c["topo"].ctypes.data_as(i32p),
                                    c["flag"].ctypes.data_as(u8p), c["ptr"].ctypes.data_as(i32p), c["node"].ctypes.data_as(i32p),
                                    c["cost"].ctypes.data_as(i64p)) == 0
    return c


def _grow(hdr, ag, group, selector, node):
    G = ag.struct.n_groups
    ptr, sel, nd = ag.array("placed_ptr"), ag.array("placed_selector"), ag.array("placed_node")
    new_sel, new_nd, new_ptr = [], [], [0]
    for g in range(G):
        extra = [j for j in range(len(group)) if group[j] == g]
        new_sel += list(sel[ptr[g]:ptr[g + 1]]) + [selector[j] for j in extra]
        new_nd += list(nd[ptr[g]:ptr[g + 1]]) + [node[j] for j in extra]
        new_ptr.append(len(new_sel))
    keep = {f: ag.array(f) for f in ("wl_ptr", "wl_selector", "dep_ptr", "dep_selector", "dep_max_cost", "topo_ptr", "topo_selector", "topo_index")}
    return Table(hdr, "spx_appgroup_objects", n_groups=G, placed_ptr=np.array(new_ptr, np.int32), placed_selector=np.array(new_sel, np.int32),
                 placed_node=np.array(new_nd, np.int32), **keep)


def test_entries_are_the_difference_of_two_flattens(hdr):
    lib = spx.lib()
    snap = synth.network_snapshot(hdr, 500, 800, seed=4, pods_per_group=25)
    pods, ag = snap["pods"], snap["appgroups"]
    rng = np.random.default_rng(8)
    G = ag.struct.n_groups
    m = 90
    group = rng.integers(-1, G, m).astype(np.int32)
    wl_ptr, wl_sel = ag.array("wl_ptr"), ag.array("wl_selector")
    selector = np.array([wl_sel[rng.integers(wl_ptr[g], wl_ptr[g + 1])] if g >= 0 else 3 for g in group], np.int32)
    node = rng.integers(-1, 500, m).astype(np.int32)
    old, new = _keys(lib, pods, ag), _keys(lib, pods, _grow(hdr, ag, group, selector, node))
    assert np.array_equal(old["pod_key"], new["pod_key"])  # the numbering depends on the pending batch only

    i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    n = C.c_int64()
    args = (pods.ref(), ag.ref(), m, group.ctypes.data_as(i32p), selector.ctypes.data_as(i32p), node.ctypes.data_as(i32p), C.byref(n))
    assert lib.spx_flatten_net_placed(*args, None, None, None) == 0
    key, nd, cost = np.zeros(n.value, np.int32), np.zeros(n.value, np.int32), np.zeros(n.value, np.int64)
    assert lib.spx_flatten_net_placed(*args, key.ctypes.data_as(i32p), nd.ctypes.data_as(i32p), cost.ctypes.data_as(i64p)) == 0
    assert n.value > 0

    # apply the entries to the old lists the way spx_update_net_placed does
    K = len(old["flag"])
    lists = [list(zip(old["node"][old["ptr"][k]:old["ptr"][k + 1]], old["cost"][old["ptr"][k]:old["ptr"][k + 1]])) for k in range(K)]
    flag = old["flag"].copy()
    for k, x, c in zip(key, nd, cost):
        if c < 0:
            flag[k] = 0 if flag[k] == 1 else flag[k]
            continue
        flag[k] = 2 if x < 0 else (0 if flag[k] == 1 else flag[k])
        lists[k].append((x, c))
    want = [list(zip(new["node"][new["ptr"][k]:new["ptr"][k + 1]], new["cost"][new["ptr"][k]:new["ptr"][k + 1]])) for k in range(K)]
    assert [sorted(l) for l in lists] == [sorted(l) for l in want]
    assert np.array_equal(flag, new["flag"])
