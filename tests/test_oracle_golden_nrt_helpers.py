"""Pins the helper-level pieces of the NRT / effective-request path — the oracle's restatements and, where the product
computes the same thing on the host (flatteners), the product — against the reference's own helper tests
(tests/golden/nrt_helpers.py)."""
import ctypes as C

import numpy as np
import pytest

from golden import nrt_helpers as G
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd.engine import Engine  # noqa: F401  (import check only; no GPU call in this file)

I64P, I32P = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
POLICY = {"none": 0, "best-effort": 1, "restricted": 2, "single-numa-node": 3}
SCOPE = {"container": 0, "pod": 1}
QOS = {"Guaranteed": 0, "Burstable": 1, "BestEffort": 2}


def _rl(cpu_mem):
    return {"cpu": f"{cpu_mem[0]}m", "memory": str(cpu_mem[1])}


def _zones(zs):
    return [{"name": f"node-{i}", "type": "Node", "resources": dict(rl)} for i, rl in zs]


def _flatten_nrt_pods(hdr, res, pods):
    """the product's host flattener (spx_flatten_nrt_slots + spx_flatten_nrt_pods) on `pods`"""
    import scheduler_plugins_amd as spx
    lib = spx.lib()
    rc = res.table(hdr)
    params = O.nrt_params(hdr, res)
    M = hdr.consts["SPX_NRT_MAX_RES"]
    slot_res, slot_flags, slot_w = np.zeros(M, np.int32), np.zeros(M, np.uint8), np.zeros(M, np.int64)
    n_res = C.c_int32()
    empty_nrt = O.build_nrt_objects(hdr, res, [None])
    assert lib.spx_flatten_nrt_slots(pods.ref(), empty_nrt.ref(), rc.ref(), params.ref(), C.byref(n_res), slot_res.ctypes.data_as(I32P),
                                     slot_flags.ctypes.data_as(C.POINTER(C.c_uint8)), slot_w.ctypes.data_as(I64P)) == 0
    R, P, CM = n_res.value, pods.struct.n_pods, hdr.consts["SPX_NRT_MAX_CTRS"]
    from scheduler_plugins_amd._abi import Table
    slots = Table(hdr, "spx_nrt_slots", n_res=R, slot_res=slot_res, slot_flags=slot_flags, slot_weight=slot_w)
    u8 = lambda n: np.zeros(n, np.uint8)
    qos, nn, nctr, kind, cpres, ppres = u8(P), u8(P), u8(P), u8(P * CM), u8(P * CM), u8(P)
    creq, preq = np.zeros(P * CM * max(R, 1), np.int64), np.zeros(P * max(R, 1), np.int64)
    U8P = C.POINTER(C.c_uint8)
    assert lib.spx_flatten_nrt_pods(pods.ref(), rc.ref(), slots.ref(), qos.ctypes.data_as(U8P), nn.ctypes.data_as(U8P),
                                    nctr.ctypes.data_as(U8P), kind.ctypes.data_as(U8P), cpres.ctypes.data_as(U8P),
                                    creq.ctypes.data_as(I64P), ppres.ctypes.data_as(U8P), preq.ctypes.data_as(I64P)) == 0
    return dict(R=R, slot_res=slot_res[:R], slot_flags=slot_flags[:R], pod_req=preq.reshape(P, max(R, 1))[:, :R], pod_present=ppres)


@pytest.mark.parametrize("case", G.EFFECTIVE_REQUEST, ids=lambda c: f"L{c[0]}")
def test_get_pod_effective_request(hdr, oracle, case):
    _, app, init, ovh, want = case
    res = O.Resources()
    pod = O.pod([O.container(_rl(r)) for r in app], [O.container(_rl(r)) for r in init], overhead=_rl(ovh) if ovh else None)
    pods = O.build_pod_objects(hdr, res, [pod])
    # oracle
    ids, qty = np.zeros(8, np.int32), np.zeros(8, np.int64)
    n = oracle.lib().orc_pod_effective_request(pods.ref(), 0, ids.ctypes.data_as(I32P), qty.ctypes.data_as(I64P), 8)
    got = {int(ids[i]): int(qty[i]) for i in range(n)}
    assert got == {res.id("cpu"): want[0], res.id("memory"): want[1]}
    # product: the pod-level request the flattener hands to the pod-scope Filter/Score
    f = _flatten_nrt_pods(hdr, res, pods)
    prod = {int(r): int(q) for r, q in zip(f["slot_res"], f["pod_req"][0])}
    assert prod == got


@pytest.mark.parametrize("case", G.INCLUDE_NON_NATIVE, ids=lambda c: f"L{c[0]}")
def test_include_non_native(hdr, oracle, case):
    """resourcerequests.IncludeNonNative (exclusive.go:28-44): a BestEffort pod is only filtered by TopologyMatch when some
    container — init and sidecar ones included — requests a non-native resource (filter.go:183-186)"""
    _, _, app, init, sidecar, want = case
    res = O.Resources()
    ctr = lambda r, sc=False: O.container(dict(r), dict(r), sidecar=sc)
    pod = O.pod([ctr(r) for r in app], [ctr(r) for r in init] + [ctr(r, True) for r in sidecar])
    pods = O.build_pod_objects(hdr, res, [pod])
    assert bool(oracle.lib().orc_include_non_native(pods.ref(), res.table(hdr).ref(), 0)) == want
    # product: the per-pod flag column the NRT sweep reads
    import scheduler_plugins_amd as spx
    from scheduler_plugins_amd._abi import Table
    lib = spx.lib()
    rc = res.table(hdr)
    M, CM = hdr.consts["SPX_NRT_MAX_RES"], hdr.consts["SPX_NRT_MAX_CTRS"]
    slot_res, slot_flags, slot_w = np.zeros(M, np.int32), np.zeros(M, np.uint8), np.zeros(M, np.int64)
    n_res = C.c_int32()
    U8P = C.POINTER(C.c_uint8)
    assert lib.spx_flatten_nrt_slots(pods.ref(), O.build_nrt_objects(hdr, res, [None]).ref(), rc.ref(), O.nrt_params(hdr, res).ref(), C.byref(n_res),
                                     slot_res.ctypes.data_as(I32P), slot_flags.ctypes.data_as(U8P), slot_w.ctypes.data_as(I64P)) == 0
    R = n_res.value
    slots = Table(hdr, "spx_nrt_slots", n_res=R, slot_res=slot_res, slot_flags=slot_flags, slot_weight=slot_w)
    u8 = lambda n: np.zeros(n, np.uint8)
    qos, nn, nctr, kind, cpres, ppres = u8(1), u8(1), u8(1), u8(CM), u8(CM), u8(1)
    creq, preq = np.zeros(CM * max(R, 1), np.int64), np.zeros(max(R, 1), np.int64)
    assert lib.spx_flatten_nrt_pods(pods.ref(), rc.ref(), slots.ref(), qos.ctypes.data_as(U8P), nn.ctypes.data_as(U8P), nctr.ctypes.data_as(U8P),
                                    kind.ctypes.data_as(U8P), cpres.ctypes.data_as(U8P), creq.ctypes.data_as(I64P), ppres.ctypes.data_as(U8P),
                                    preq.ctypes.data_as(I64P)) == 0
    assert bool(nn[0]) == want


@pytest.mark.parametrize("name", sorted(G.RESOURCE_CLASSES))
def test_resource_classes(hdr, oracle, name):
    host_level, affine = G.RESOURCE_CLASSES[name]
    res = O.Resources()
    rid = res.id(name)
    rc = res.table(hdr)
    assert bool(oracle.lib().orc_nrt_is_host_level(rc.ref(), rid)) == host_level
    assert bool(oracle.lib().orc_nrt_is_numa_affine(rc.ref(), rid)) == affine
    # product: the slot flags the kernels branch on
    pods = O.build_pod_objects(hdr, res, [O.pod([O.container({name: "1"})])])
    f = _flatten_nrt_pods(hdr, res, pods)
    flags = int(f["slot_flags"][list(f["slot_res"]).index(rid)])
    assert bool(flags & hdr.consts["SPX_NRT_SLOT_HOST_LEVEL"]) == host_level
    assert bool(flags & hdr.consts["SPX_NRT_SLOT_AFFINE"]) == affine


def _zone_state(hdr, res, zones_after, names):
    return [[res.canonical(n, rl[n]) if n in rl else -1 for n in names] for _, rl in zones_after]


@pytest.mark.parametrize("case", G.SUBTRACT_NUMA, ids=lambda c: f"L{c['line']}")
def test_subtract_resources_from_numa_node_list(hdr, oracle, case):
    res = O.Resources()
    nrts = O.build_nrt_objects(hdr, res, [O.nrt(_zones(case["zones"]))])
    pods = O.build_pod_objects(hdr, res, [O.pod([O.container(case["request"])])])
    names = sorted({n for _, rl in case["zones"] for n in rl} | set(case["request"]))
    q = np.array([res.id(n) for n in names] or [0], dtype=np.int32)
    out = np.full(len(case["zones"]) * len(q), -7, np.int64)
    rc = oracle.lib().orc_nrt_test_subtract_numa(nrts.ref(), res.table(hdr).ref(), 0, case["numa_id"], QOS[case["qos"]], pods.ref(), 0,
                                                 q.ctypes.data_as(I32P), len(names), out.ctypes.data_as(I64P))
    if case["expected"] is None:
        assert rc != 0
        return
    assert rc == 0
    if names:
        assert out.reshape(len(case["zones"]), -1).tolist() == _zone_state(hdr, res, case["expected"], names)


@pytest.mark.parametrize("case", G.SUBTRACT_NUMAS, ids=lambda c: f"L{c['line']}")
def test_subtract_from_numas(hdr, oracle, case):
    res = O.Resources()
    nrts = O.build_nrt_objects(hdr, res, [O.nrt(_zones(case["zones"]))])
    pods = O.build_pod_objects(hdr, res, [O.pod([O.container(case["request"])])])
    names = sorted(case["request"])
    q = np.array([res.id(n) for n in names], dtype=np.int32)
    out = np.zeros(len(case["zones"]) * len(q), np.int64)
    bits = sum(1 << i for i in case["nodes"])
    oracle.lib().orc_nrt_test_subtract_numas(nrts.ref(), 0, pods.ref(), 0, C.c_uint64(bits), q.ctypes.data_as(I32P), len(names),
                                             out.ctypes.data_as(I64P))
    assert out.reshape(len(case["zones"]), -1).tolist() == _zone_state(hdr, res, case["expected"], names)


@pytest.mark.parametrize("case", G.ONLY_NON_NUMA, ids=lambda c: f"L{c[0]}")
def test_only_non_numa_resources(hdr, oracle, case):
    _, request, want = case
    res = O.Resources()
    nrts = O.build_nrt_objects(hdr, res, [O.nrt(_zones(G.ONLY_NON_NUMA_ZONES))])
    pods = O.build_pod_objects(hdr, res, [O.pod([O.container(request)])])
    assert bool(oracle.lib().orc_nrt_only_non_numa(nrts.ref(), 0, pods.ref(), 0)) == want


def _flatten_nrt_node_conf(hdr, res, nrts):
    """(single-numa-node?, pod scope?, MaxNUMANodes) as the product's node flattener encodes them"""
    import scheduler_plugins_amd as spx
    from scheduler_plugins_amd._abi import Table
    lib = spx.lib()
    nodes = O.build_node_objects(hdr, res, [O.node({"cpu": "1"})])
    slots = Table(hdr, "spx_nrt_slots", n_res=0, slot_res=np.zeros(8, np.int32), slot_flags=np.zeros(8, np.uint8),
                  slot_weight=np.zeros(8, np.int64))
    Z = hdr.consts["SPX_NRT_MAX_ZONES"]
    U8P = C.POINTER(C.c_uint8)
    flags, mx, nz, zid, zp, npres = (np.zeros(1, np.uint8), np.zeros(1, np.int32), np.zeros(1, np.uint8), np.zeros(Z, np.uint8),
                                     np.zeros(Z, np.uint8), np.zeros(1, np.uint8))
    avail, cost, mind = np.zeros(Z * 1, np.int64), np.zeros(Z * Z, np.int32), np.zeros(Z, np.float32)
    fn = lib.spx_flatten_nrt_nodes
    args = [nodes.ref(), nrts.ref(), slots.ref()]
    outs = [flags, mx, nz, zid, zp, avail, cost, mind, npres]
    assert fn(*args, *[o.ctypes.data_as(t) for o, t in zip(outs, fn.argtypes[3:])]) == 0
    f = int(flags[0])
    return bool(f & hdr.consts["SPX_NRT_F_SINGLE_NUMA"]), bool(f & hdr.consts["SPX_NRT_F_POD_SCOPE"]), int(mx[0])


def _check_conf(hdr, oracle, policies, attributes, want):
    res = O.Resources()
    nrts = O.build_nrt_objects(hdr, res, [O.nrt([], policies, attributes)])
    p, s, m = C.c_int(), C.c_int(), C.c_int()
    oracle.lib().orc_nrt_conf(nrts.ref(), 0, C.byref(p), C.byref(s), C.byref(m))
    assert (p.value, s.value, m.value) == (POLICY[want[0]], SCOPE[want[1]], want[2])
    single, pod_scope, max_numa = _flatten_nrt_node_conf(hdr, res, nrts)
    assert (single, pod_scope, max_numa) == (want[0] == "single-numa-node", want[1] == "pod", want[2])


@pytest.mark.parametrize("case", G.CONFIG_FROM_NRT, ids=lambda c: f"L{c[0]}")
def test_config_from_nrt(hdr, oracle, case):
    _check_conf(hdr, oracle, case[1], case[2], case[3])


@pytest.mark.parametrize("case", G.CONFIG_FROM_ATTRIBUTES, ids=lambda c: f"L{c[0]}")
def test_config_from_attributes(hdr, oracle, case):
    _check_conf(hdr, oracle, [], case[1], case[2])


@pytest.mark.parametrize("case", G.CONFIG_FROM_POLICIES, ids=lambda c: f"L{c[0]}")
def test_config_from_policies(hdr, oracle, case):
    _check_conf(hdr, oracle, case[1], {}, case[2])


@pytest.mark.parametrize("case", G.OVER_RESERVE, ids=lambda c: f"{c['source']}:{c['line']}")
def test_over_reserve_assumed_pods(hdr, oracle, case):
    """N11: the zone table the kernels read = NRT Available - assumed pods' effective requests (floored at 0)"""
    import scheduler_plugins_amd as spx
    from scheduler_plugins_amd._abi import Table
    res = O.Resources()
    assumed = []
    for ctrs in case["assumed_pods"]:  # what resourceStore.AddPod records: util.GetPodEffectiveRequest(pod)
        pods = O.build_pod_objects(hdr, res, [O.pod([O.container(c) for c in ctrs])])
        ids, qty = np.zeros(8, np.int32), np.zeros(8, np.int64)
        n = oracle.lib().orc_pod_effective_request(pods.ref(), 0, ids.ctypes.data_as(I32P), qty.ctypes.data_as(I64P), 8)
        names = {res.id(k): k for c in ctrs for k in c}
        assumed.append({names[int(ids[i])]: (f"{int(qty[i])}m" if names[int(ids[i])] == "cpu" else str(int(qty[i]))) for i in range(n)})
    nrts = O.build_nrt_objects(hdr, res, [O.nrt(_zones(case["zones"]), ["SingleNUMANodeContainerLevel"])], assumed={0: assumed})
    names = sorted({n for _, rl in case["zones"] for n in rl})
    want = _zone_state(hdr, res, case["expected"], names)
    # oracle: the NUMA list the Filter/Score restatement works on (an empty request subtracts nothing)
    empty = O.build_pod_objects(hdr, res, [O.pod([O.container({})])])
    q = np.array([res.id(n) for n in names], dtype=np.int32)
    out = np.zeros(len(case["zones"]) * len(q), np.int64)
    assert oracle.lib().orc_nrt_test_subtract_numa(nrts.ref(), res.table(hdr).ref(), 0, 0, 0, empty.ref(), 0, q.ctypes.data_as(I32P),
                                                   len(names), out.ctypes.data_as(I64P)) == 0
    assert out.reshape(len(case["zones"]), -1).tolist() == want
    # product: spx_flatten_nrt_nodes' zone_avail for slots (cpu, memory, nic)
    lib = spx.lib()
    R = len(names)
    slot_res = np.zeros(8, np.int32)
    slot_res[:R] = q
    slots = Table(hdr, "spx_nrt_slots", n_res=R, slot_res=slot_res, slot_flags=np.zeros(8, np.uint8), slot_weight=np.ones(8, np.int64))
    nodes = O.build_node_objects(hdr, res, [O.node_from_zones(_zones(case["zones"]))])
    Z = hdr.consts["SPX_NRT_MAX_ZONES"]
    outs = [np.zeros(1, np.uint8), np.zeros(1, np.int32), np.zeros(1, np.uint8), np.zeros(Z, np.uint8), np.zeros(Z, np.uint8),
            np.zeros(Z * R, np.int64), np.zeros(Z * Z, np.int32), np.zeros(Z, np.float32), np.zeros(1, np.uint8)]
    fn = lib.spx_flatten_nrt_nodes
    assert fn(nodes.ref(), nrts.ref(), slots.ref(), *[o.ctypes.data_as(t) for o, t in zip(outs, fn.argtypes[3:])]) == 0
    avail, present = outs[5].reshape(Z, R), outs[4]
    got = [[int(avail[z][r]) if (present[z] >> r) & 1 else -1 for r in range(R)] for z in range(len(case["zones"]))]
    assert got == want


@pytest.mark.parametrize("case", G.MIN_DISTANCE, ids=lambda c: f"L{c[0]}")
def test_min_avg_distance_in_combinations(hdr, case):
    """the per-node, per-subset-size minimal average distance the LeastNUMANodes kernels compare against
    (spx_flatten_nrt_nodes' min_avg_dist), float32 like the reference"""
    import scheduler_plugins_amd as spx
    from scheduler_plugins_amd._abi import Table
    _, with_costs, k, want = case
    res = O.Resources()
    zones = [{"name": f"node-{i}", "type": "Node", "resources": {"cpu": "4"},
              "costs": ({f"node-{j}": c for j, c in G.MIN_DISTANCE_COSTS[i].items()} if with_costs else {})} for i in range(4)]
    nrts = O.build_nrt_objects(hdr, res, [O.nrt(zones, ["SingleNUMANodePodLevel"])])
    nodes = O.build_node_objects(hdr, res, [O.node_from_zones(zones)])
    slot_res = np.zeros(8, np.int32)
    slot_res[0] = res.id("cpu")
    slots = Table(hdr, "spx_nrt_slots", n_res=1, slot_res=slot_res, slot_flags=np.zeros(8, np.uint8), slot_weight=np.ones(8, np.int64))
    Z = hdr.consts["SPX_NRT_MAX_ZONES"]
    outs = [np.zeros(1, np.uint8), np.zeros(1, np.int32), np.zeros(1, np.uint8), np.zeros(Z, np.uint8), np.zeros(Z, np.uint8),
            np.zeros(Z, np.int64), np.zeros(Z * Z, np.int32), np.zeros(Z, np.float32), np.zeros(1, np.uint8)]
    fn = spx.lib().spx_flatten_nrt_nodes
    assert fn(nodes.ref(), nrts.ref(), slots.ref(), *[o.ctypes.data_as(t) for o, t in zip(outs, fn.argtypes[3:])]) == 0
    assert outs[7][k - 1] == np.float32(want)
