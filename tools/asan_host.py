"""Runs every host-side entry point of the library (flatteners, comparator, quota, LROC/Peaks columns) on synthetic snapshots of
four sizes under AddressSanitizer + UBSan.  The host sources are compiled on their own (no HIP needed):

    mkdir -p /tmp/asan && g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -pthread -Iinclude -shared -fPIC \\
        scheduler-plugins_amd/host/*.cc -o /tmp/asan/libhost_asan.so
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_host.py

Last run: clean (round 1).  The wire-format decoder is fuzzed the same way by tests/test_ingest_nrt.py::test_decoder_survives_mutated_input."""
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
print("host asan ok")
