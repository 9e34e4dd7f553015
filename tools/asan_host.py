"""Runs every host-side entry point of the library (flatteners, comparator, quota, LROC/Peaks columns) on synthetic snapshots of
four sizes under AddressSanitizer + UBSan.  The host sources are compiled on their own (no HIP needed):

    mkdir -p /tmp/asan && g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -pthread -Iinclude -shared -fPIC \\
        scheduler-plugins_amd/host/*.cc -o /tmp/asan/libhost_asan.so
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_host.py

Last run: clean (round 5, including the decoder fuzz at the end; before that round 2).  The wire-format decoder is fuzzed the same way by tests/test_ingest_nrt.py::test_decoder_survives_mutated_input."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
import scheduler_plugins_amd as spx
from scheduler_plugins_amd import synth, objects as O
from scheduler_plugins_amd._abi import Table
hdr = spx.header()
L = C.CDLL("/tmp/asan/libhost_asan.so")
missing = hdr.bind(L, [n for n in hdr.protos if n.startswith("spx_flatten") or n.startswith("spx_toposort") or n.startswith("spx_nrt_post") or n.startswith("spx_ingest")])
print("unbound:", missing)
i64p, i32p, u8p, f64p, f32p = (C.POINTER(t) for t in (C.c_int64, C.c_int32, C.c_uint8, C.c_double, C.c_float))
def outs(fn, skip, sizes):
    arrs = []
    for t, n in zip(fn.argtypes[skip:], sizes):
        dt = {i64p: np.int64, i32p: np.int32, u8p: np.uint8, f64p: np.float64, f32p: np.float32}[t]
        arrs.append(np.zeros(max(n, 1), dt))
    return arrs
for seed, (N, P) in enumerate([(1, 1), (7, 3), (300, 129), (5000, 800)]):
    snap = synth.full_snapshot(hdr, N, P, seed=seed + 1)
    tlp = Table(hdr, "spx_tlp_params", target_utilization=40, default_requests_milli=1000, requests_multiplier=1.5)
    ap = Table(hdr, "spx_allocatable_params", mode=0, n_res=2, res=np.array([1, 0], np.int32), weight=np.array([1, 1 << 20], np.int64))
    a = np.zeros(2 * N, np.int64)
    assert L.spx_flatten_alloc_nodes(snap["nodes"].ref(), snap["rc"].ref(), ap.ref(), a.ctypes.data_as(i64p)) == 0
    fn = L.spx_flatten_trimaran_nodes
    o = outs(fn, 4, [N] * 11)
    assert fn(snap["nodes"].ref(), snap["metrics"].ref(), snap["assigned"].ref(), tlp.ref(), *[x.ctypes.data_as(t) for x, t in zip(o, fn.argtypes[4:])]) == 0
    fn = L.spx_flatten_trimaran_pods
    o = outs(fn, 2, [P] * 3)
    assert fn(snap["pods"].ref(), tlp.ref(), *[x.ctypes.data_as(t) for x, t in zip(o, fn.argtypes[2:])]) == 0
    # lroc / peaks
    np_ = synth.synth_node_pods(hdr, N, seed)
    fn = L.spx_flatten_lroc_nodes
    o = outs(fn, 2, [N] * 4)
    assert fn(snap["nodes"].ref(), np_.ref(), *[x.ctypes.data_as(t) for x, t in zip(o, fn.argtypes[2:])]) == 0
    fn = L.spx_flatten_lroc_pods
    o = outs(fn, 1, [P] * 4)
    assert fn(snap["pods"].ref(), *[x.ctypes.data_as(t) for x, t in zip(o, fn.argtypes[1:])]) == 0
    pm = synth.synth_power_models(hdr, N, seed)
    fn = L.spx_flatten_peaks_nodes
    o = outs(fn, 3, [N] * 5)
    assert fn(snap["nodes"].ref(), snap["metrics"].ref(), pm.ref(), *[x.ctypes.data_as(t) for x, t in zip(o, fn.argtypes[3:])]) == 0
    cpu = np.zeros(P, np.int64)
    assert L.spx_flatten_peaks_pods(snap["pods"].ref(), cpu.ctypes.data_as(i64p)) == 0
    # nrt
    params = O.nrt_params(hdr, O.Resources(), "LeastAllocated")
    n_res = C.c_int32()
    sr, sf, sw = np.zeros(8, np.int32), np.zeros(8, np.uint8), np.zeros(8, np.int64)
    assert L.spx_flatten_nrt_slots(snap["pods"].ref(), snap["nrt"].ref(), snap["rc"].ref(), params.ref(), C.byref(n_res), sr.ctypes.data_as(i32p), sf.ctypes.data_as(u8p), sw.ctypes.data_as(i64p)) == 0
    R = n_res.value
    slots = Table(hdr, "spx_nrt_slots", n_res=R, slot_res=sr, slot_flags=sf, slot_weight=sw)
    fn = L.spx_flatten_nrt_nodes
    o = outs(fn, 3, [N, N, N, N * 8, N * 8, N * 8 * max(R, 1), N * 64, N * 8, N])
    assert fn(snap["nodes"].ref(), snap["nrt"].ref(), slots.ref(), *[x.ctypes.data_as(t) for x, t in zip(o, fn.argtypes[3:])]) == 0
    fn = L.spx_flatten_nrt_pods
    o = outs(fn, 3, [P, P, P, P * 8, P * 8, P * 8 * max(R, 1), P, P * max(R, 1)])
    assert fn(snap["pods"].ref(), snap["rc"].ref(), slots.ref(), *[x.ctypes.data_as(t) for x, t in zip(o, fn.argtypes[3:])]) == 0
    # network
    nt = snap["nettopo"].struct
    rcost, zcost = np.zeros(max(nt.n_regions ** 2, 1), np.int32), np.zeros(max(nt.n_zones ** 2, 1), np.int32)
    assert L.spx_flatten_net_topo(snap["nettopo"].ref(), rcost.ctypes.data_as(i32p), zcost.ctypes.data_as(i32p)) == 0
    nk, npairs = C.c_int32(), C.c_int64()
    assert L.spx_flatten_net_keys(snap["pods"].ref(), snap["appgroups"].ref(), C.byref(nk), C.byref(npairs), None, None, None, None, None, None) == 0
    pk, to = np.zeros(P, np.int32), np.zeros(P, np.int32)
    kse, pp = np.zeros(max(nk.value, 1), np.uint8), np.zeros(nk.value + 1, np.int32)
    pn, pmx = np.zeros(max(npairs.value, 1), np.int32), np.zeros(max(npairs.value, 1), np.int64)
    assert L.spx_flatten_net_keys(snap["pods"].ref(), snap["appgroups"].ref(), C.byref(nk), C.byref(npairs), pk.ctypes.data_as(i32p), to.ctypes.data_as(i32p), kse.ctypes.data_as(u8p), pp.ctypes.data_as(i32p), pn.ctypes.data_as(i32p), pmx.ctypes.data_as(i64p)) == 0
    ne = C.c_int64()
    assert L.spx_flatten_net_commit(snap["pods"].ref(), snap["appgroups"].ref(), C.byref(ne), None, None, None) == 0
    ep, ek, ec = np.zeros(P + 1, np.int32), np.zeros(max(ne.value, 1), np.int32), np.zeros(max(ne.value, 1), np.int64)
    assert L.spx_flatten_net_commit(snap["pods"].ref(), snap["appgroups"].ref(), C.byref(ne), ep.ctypes.data_as(i32p), ek.ctypes.data_as(i32p), ec.ctypes.data_as(i64p)) == 0
    assert ep[-1] == ne.value
    a_, b_ = np.arange(P, dtype=np.int64), np.arange(P, dtype=np.int64)[::-1].copy()
    out = np.zeros(P, np.uint8)
    assert L.spx_toposort_less(snap["pods"].ref(), to.ctypes.data_as(i32p), P, a_.ctypes.data_as(i64p), b_.ctypes.data_as(i64p), out.ctypes.data_as(u8p)) == 0
    # quota
    q = snap["quota"].struct
    NS = q.n_namespaces
    fn = L.spx_flatten_quota
    o = outs(fn, 3, [P, P, P * 8, P, 8, 1, 8, 1, NS * 8, NS, NS + 1, max(q.n_nominated, 1), max(q.n_nominated, 1), max(q.n_nominated, 1) * 8, max(q.n_nominated, 1)])
    assert fn(snap["pods"].ref(), snap["rc"].ref(), snap["quota"].ref(), *[x.ctypes.data_as(t) for x, t in zip(o, fn.argtypes[3:])]) == 0
    print("ok", N, P)

# wire-format decoders added in round 2 (load-watcher metrics; AppGroups / pods in any order with selector renumbering; nominated
# pods): valid documents, then truncated / byte-flipped / spliced mutations — the decoder must answer with an error code or a
# valid table, never touch memory it does not own
import json, random
random.seed(7)
names = [f"n{i}" for i in range(6)]
pp_ = C.POINTER(C.POINTER(C.c_char))
def mk():
    h = C.POINTER(hdr.opaque["spx_ingest"])()
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    assert L.spx_ingest_create(C.cast(arr, pp_), len(names), None, 0, C.byref(h)) == 0
    return h
metrics = json.dumps({"timestamp": 1, "window": {"duration": "15m", "start": 1, "end": 2}, "source": "t", "data": {"NodeMetricsMap": {
    n: {"metrics": [{"name": "x", "type": t, "operator": o, "value": v} for t, o, v in (("CPU", "AVG", 10.5), ("Memory", "STD", 3), ("CPU", "", 1e-3))], "tags": {}, "metadata": {}}
    for n in names + ["other"]}}}).encode()
wl = lambda s_: {"selector": s_}
groups = json.dumps([{"metadata": {"name": g}, "spec": {"workloads": [{"workload": wl(a), "dependencies": [{"workload": wl(b), "maxNetworkCost": 5}]} for a, b in (("zz", "aa"), ("mm", "zz"))]},
                      "status": {"topologyOrder": [{"workload": wl("aa"), "index": 1}, {"workload": wl("zz"), "index": 2}]}} for g in ("g1", "g2")]).encode()
pods = json.dumps([{"metadata": {"namespace": "ns", "labels": {"appgroup.diktyo.x-k8s.io": g, "appgroup.diktyo.x-k8s.io.workload": s_}},
                    "spec": {"priority": 3, "containers": [{"name": "c", "resources": {"requests": {"cpu": "100m"}}}]}, "status": {"nominatedNodeName": nn}}
                   for g, s_, nn in (("g2", "zz", "n1"), ("g9", "00", "nowhere"), ("g1", "mm", "n0"))]).encode()
quota = json.dumps([{"metadata": {"namespace": "ns"}, "spec": {"min": {"cpu": "1"}, "max": {"cpu": "2"}}, "status": {"used": {"cpu": "500m"}}}]).encode()
def mutate(b):
    b = bytearray(b)
    k = random.random()
    if k < 0.4: return bytes(b[: random.randrange(len(b))])
    if k < 0.8:
        for _ in range(random.randrange(1, 4)): b[random.randrange(len(b))] = random.randrange(256)
        return bytes(b)
    i, j = sorted(random.randrange(len(b)) for _ in range(2))
    return bytes(b[:i] + b[j:] + b[i:j])
n64, u64 = C.c_int64(), C.c_int64()
for it in range(400):
    h = mk()
    docs = [("m", metrics), ("g", groups), ("p", pods), ("q", quota)]
    random.shuffle(docs)
    for kind, d in docs + docs[:2]:
        d = d if it == 0 or random.random() < 0.3 else mutate(d)
        if kind == "m": L.spx_ingest_metrics_json(h, d, len(d), C.byref(n64), C.byref(u64))
        elif kind == "g": L.spx_ingest_appgroups_json(h, d, len(d), C.byref(n64))
        elif kind == "p": L.spx_ingest_pods_json(h, d, len(d), C.byref(n64))
        else:
            ns = (C.c_char_p * 1)(b"ns")
            L.spx_ingest_quota_json(h, d, len(d), C.cast(ns, pp_), 1, C.byref(n64), C.byref(u64))
        # walk the tables the way a consumer would
        t = L.spx_ingest_appgroup_objects(h).contents
        for g in range(t.n_groups):
            for w in range(t.wl_ptr[g], t.wl_ptr[g + 1]):
                assert t.wl_selector[w] >= 0
                for k in range(t.dep_ptr[w], t.dep_ptr[w + 1]): assert t.dep_selector[k] >= 0
        pt = L.spx_ingest_pod_objects(h).contents
        for i in range(pt.n_pods): assert pt.selector[i] >= -1 and pt.appgroup[i] < max(t.n_groups, 1) + 100
        mp = L.spx_ingest_metrics_objects(h)
        if mp:
            m = mp.contents
            assert m.m_ptr[len(names)] >= 0 and sum(m.node_present[i] for i in range(len(names))) <= len(names)
        qp = L.spx_ingest_quota_objects(h)
        if qp:
            q_ = qp.contents
            for j in range(q_.n_nominated): assert 0 <= q_.nom_pending_index[j] < pt.n_pods
    L.spx_ingest_destroy(h)
print("decoder fuzz ok")
print("host asan ok")
