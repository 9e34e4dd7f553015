"""spx_flatten_net_commit (host/flatten_network.cc): what binding pod p adds to NetworkOverhead's view of the pods scheduled after
it — every workload key of p's AppGroup that has dependencies stops scoring equally (entry (key, -1)), and every dependency of
such a workload on p's selector gains a (host, MaxNetworkCost) pair (entry (key, cost)); networkoverhead.go:174-298 through
util.GetScheduledList.  The C function computes the list once per (AppGroup, selector) and copies it per pod; this test
restates the definition per pod in Python and compares the CSR, and checks the key numbering against spx_flatten_net_keys."""
import ctypes as C

import numpy as np

import scheduler_plugins_amd as spx
from scheduler_plugins_amd import synth


def _commit(lib, pods, ag):
    i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    n = C.c_int64()
    assert lib.spx_flatten_net_commit(pods.ref(), ag.ref(), C.byref(n), None, None, None) == 0
    P = pods.struct.n_pods
    ptr, key, cost = np.zeros(P + 1, np.int32), np.zeros(max(n.value, 1), np.int32), np.zeros(max(n.value, 1), np.int64)
    assert lib.spx_flatten_net_commit(pods.ref(), ag.ref(), C.byref(n), ptr.ctypes.data_as(i32p), key.ctypes.data_as(i32p), cost.ctypes.data_as(i64p)) == 0
    return n.value, ptr, key, cost


def _keys(lib, pods, ag):
    i32p, i64p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
    nk, npairs = C.c_int32(), C.c_int64()
    assert lib.spx_flatten_net_keys(pods.ref(), ag.ref(), C.byref(nk), C.byref(npairs), None, None, None, None, None, None) == 0
    P = pods.struct.n_pods
    pod_key, topo = np.zeros(P, np.int32), np.zeros(P, np.int32)
    se, pp = np.zeros(nk.value, np.uint8), np.zeros(nk.value + 1, np.int32)
    pn, pc = np.zeros(max(npairs.value, 1), np.int32), np.zeros(max(npairs.value, 1), np.int64)
    assert lib.spx_flatten_net_keys(pods.ref(), ag.ref(), C.byref(nk), C.byref(npairs), pod_key.ctypes.data_as(i32p), topo.ctypes.data_as(i32p),
                                    se.ctypes.data_as(u8p), pp.ctypes.data_as(i32p), pn.ctypes.data_as(i32p), pc.ctypes.data_as(i64p)) == 0
    return pod_key


def test_commit_effects_match_the_per_pod_definition():
    hdr, lib = spx.header(), spx.lib()
    for seed, n_nodes, n_pods in ((3, 50, 400), (4, 200, 3000)):
        snap = synth.network_snapshot(hdr, n_nodes, n_pods, seed=seed)
        pods, ag = snap["pods"], snap["appgroups"]
        n, ptr, key, cost = _commit(lib, pods, ag)
        pod_key = _keys(lib, pods, ag)
        group, sel = pods.array("appgroup"), pods.array("selector")
        wl_ptr, wl_sel = ag.array("wl_ptr"), ag.array("wl_selector")
        dep_ptr, dep_sel, dep_max = ag.array("dep_ptr"), ag.array("dep_selector"), ag.array("dep_max_cost")
        G = ag.struct.n_groups
        # key of every (group, selector) that occurs among the pods, numbered as spx_flatten_net_keys numbers them, and per group
        # the selectors in order of first appearance
        key_of, by_group = {}, {}
        for p in range(n_pods):
            g = int(group[p])
            if 0 <= g < G and (g, int(sel[p])) not in key_of:
                key_of[(g, int(sel[p]))] = int(pod_key[p])
                by_group.setdefault(g, []).append(int(sel[p]))
        want_ptr, want = [0], []
        for p in range(n_pods):
            g, s = int(group[p]), int(sel[p])
            if 0 <= g < G:
                for ks in by_group[g]:
                    wls = [w for w in range(wl_ptr[g], wl_ptr[g + 1]) if wl_sel[w] == ks]
                    if not any(dep_ptr[w + 1] > dep_ptr[w] for w in wls):
                        continue
                    want.append((key_of[(g, ks)], -1))
                    for w in wls:
                        for d in range(dep_ptr[w], dep_ptr[w + 1]):
                            if dep_sel[d] == s:
                                want.append((key_of[(g, ks)], int(dep_max[d])))
            want_ptr.append(len(want))
        assert n == len(want) and n > 0
        assert np.array_equal(ptr, np.array(want_ptr, np.int32))
        assert np.array_equal(key[:n], np.array([k for k, _ in want], np.int32))
        assert np.array_equal(cost[:n], np.array([c for _, c in want], np.int64))


def test_sizing_result_is_reused_once_and_only_for_the_same_tables():
    """Both flatteners run ONE pass: the sizing call (NULL arrays) leaves its result for the fill call that follows it with the
    same tables, per calling thread; a fill call for other tables (or a second fill) computes afresh.  Whatever the sequence, the
    arrays are those of a fresh computation."""
    hdr, lib = spx.header(), spx.lib()
    a = synth.network_snapshot(hdr, 40, 300, seed=5)
    b = synth.network_snapshot(hdr, 40, 300, seed=6)      # same sizes, other contents
    want_a, want_b = _commit(lib, a["pods"], a["appgroups"]), _commit(lib, b["pods"], b["appgroups"])
    assert want_a[0] != want_b[0] or not np.array_equal(want_a[2], want_b[2])
    i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)

    def fill(snap, n_alloc):
        P = snap["pods"].struct.n_pods
        n = C.c_int64()
        ptr, key, cost = np.zeros(P + 1, np.int32), np.zeros(max(n_alloc, 1), np.int32), np.zeros(max(n_alloc, 1), np.int64)
        assert lib.spx_flatten_net_commit(snap["pods"].ref(), snap["appgroups"].ref(), C.byref(n), ptr.ctypes.data_as(i32p),
                                          key.ctypes.data_as(i32p), cost.ctypes.data_as(i64p)) == 0
        return n.value, ptr, key, cost

    # size A, then fill B (other tables): B's arrays, not A's leftovers
    n = C.c_int64()
    assert lib.spx_flatten_net_commit(a["pods"].ref(), a["appgroups"].ref(), C.byref(n), None, None, None) == 0
    got = fill(b, max(want_a[0], want_b[0]))
    assert got[0] == want_b[0] and np.array_equal(got[1], want_b[1]) and np.array_equal(got[2][:got[0]], want_b[2][:got[0]])
    # two fills in a row without a sizing call in between
    for _ in range(2):
        got = fill(a, want_a[0])
        assert got[0] == want_a[0] and np.array_equal(got[1], want_a[1]) and np.array_equal(got[3][:got[0]], want_a[3][:got[0]])
    # the keys flattener: the same protocol
    ka, kb = _keys(lib, a["pods"], a["appgroups"]), _keys(lib, b["pods"], b["appgroups"])
    nk, npairs = C.c_int32(), C.c_int64()
    assert lib.spx_flatten_net_keys(a["pods"].ref(), a["appgroups"].ref(), C.byref(nk), C.byref(npairs), None, None, None, None, None, None) == 0
    assert np.array_equal(_keys(lib, b["pods"], b["appgroups"]), kb) and np.array_equal(_keys(lib, a["pods"], a["appgroups"]), ka)


def test_sizing_result_is_not_served_to_tables_that_changed_in_place():
    """The advisor's scenario (round 3): an ingest handle hands out the SAME table addresses every cycle and the batch size repeats.
    A sizing call of cycle 1 that is never followed by its fill call (or whose fill lands on another OS thread under cgo) must not be
    served to cycle 2's fill call: the cache is keyed on the tables' content, so contents changed in place count as other tables."""
    hdr, lib = spx.header(), spx.lib()
    a = synth.network_snapshot(hdr, 40, 300, seed=5)
    i32p, i64p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
    P = a["pods"].struct.n_pods
    n = C.c_int64()
    nk, npairs = C.c_int32(), C.c_int64()
    # cycle 1: sizing calls, never filled
    assert lib.spx_flatten_net_commit(a["pods"].ref(), a["appgroups"].ref(), C.byref(n), None, None, None) == 0
    assert lib.spx_flatten_net_keys(a["pods"].ref(), a["appgroups"].ref(), C.byref(nk), C.byref(npairs), None, None, None, None, None, None) == 0
    old_key = _keys(lib, a["pods"], a["appgroups"]).copy()
    assert lib.spx_flatten_net_keys(a["pods"].ref(), a["appgroups"].ref(), C.byref(nk), C.byref(npairs), None, None, None, None, None, None) == 0
    # cycle 2: the same table objects (same addresses, same n_pods) hold another batch
    for name in ("appgroup", "selector"):
        col = a["pods"].array(name)
        col[:] = np.roll(col, 7)
    # fill-only calls, as a caller whose sizing call ran on another OS thread would issue them (room for any result)
    room = 8 * max(n.value, 1) + 64
    ptr, key, cost = np.zeros(P + 1, np.int32), np.zeros(room, np.int32), np.zeros(room, np.int64)
    got_n = C.c_int64()
    assert lib.spx_flatten_net_commit(a["pods"].ref(), a["appgroups"].ref(), C.byref(got_n), ptr.ctypes.data_as(i32p), key.ctypes.data_as(i32p),
                                      cost.ctypes.data_as(i64p)) == 0
    pod_key, topo = np.zeros(P, np.int32), np.zeros(P, np.int32)
    se, pp = np.zeros(P + 1, np.uint8), np.zeros(P + 2, np.int32)
    pn, pc = np.zeros(8 * max(npairs.value, 1) + 64, np.int32), np.zeros(8 * max(npairs.value, 1) + 64, np.int64)
    assert lib.spx_flatten_net_keys(a["pods"].ref(), a["appgroups"].ref(), C.byref(nk), C.byref(npairs), pod_key.ctypes.data_as(i32p), topo.ctypes.data_as(i32p),
                                    se.ctypes.data_as(u8p), pp.ctypes.data_as(i32p), pn.ctypes.data_as(i32p), pc.ctypes.data_as(i64p)) == 0
    want = _commit(lib, a["pods"], a["appgroups"])          # sizing + fill: a fresh computation on the current contents
    want_key = _keys(lib, a["pods"], a["appgroups"])
    assert got_n.value == want[0] and np.array_equal(ptr, want[1]) and np.array_equal(key[:got_n.value], want[2][:got_n.value])
    assert np.array_equal(pod_key, want_key) and not np.array_equal(pod_key, old_key)
