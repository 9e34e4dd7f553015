"""Host flatteners of LowRiskOverCommitment (scheduler-plugins_amd/host/flatten_lroc.cc) against the oracle's
GetResourceRequested / GetResourceLimits / GetNodeRequestsAndLimits restatement.  CPU only."""
import ctypes as C

import numpy as np

import scheduler_plugins_amd as spx
from golden import lroc as GL
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd import synth

I64P = C.POINTER(C.c_int64)
COLS = ("req_cpu_milli", "req_mem", "lim_cpu_milli", "lim_mem")


def flatten_pods(pods):
    cols = {k: np.zeros(pods.struct.n_pods, np.int64) for k in COLS}
    assert spx.lib().spx_flatten_lroc_pods(pods.ref(), *[v.ctypes.data_as(I64P) for v in cols.values()]) == 0
    return cols


def flatten_nodes(nodes, node_pods):
    cols = {k: np.zeros(nodes.struct.n_nodes, np.int64) for k in COLS}
    assert spx.lib().spx_flatten_lroc_nodes(nodes.ref(), node_pods.ref() if node_pods else None, *[v.ctypes.data_as(I64P) for v in cols.values()]) == 0
    return cols


def oracle_pod(oracle, pods, i):
    v = [C.c_int64() for _ in range(4)]
    oracle.lib().orc_get_resource_requested(pods.ref(), i, C.byref(v[0]), C.byref(v[1]))
    oracle.lib().orc_get_resource_limits(pods.ref(), i, C.byref(v[2]), C.byref(v[3]))
    r = [x.value for x in v]
    return [r[0], r[1], max(r[2], r[0]), max(r[3], r[1])]


def test_pods_match_oracle(hdr, oracle):
    pods = synth.synth_pods(hdr, 3000, seed=5)
    cols = flatten_pods(pods)
    for i in range(3000):
        assert [int(cols[k][i]) for k in COLS] == oracle_pod(oracle, pods, i)
    assert (cols["lim_cpu_milli"] > cols["req_cpu_milli"]).any() and ((cols["req_cpu_milli"] == 0) & (cols["lim_mem"] == 0)).any()


def test_reference_pod_fixtures(hdr):
    res = O.Resources()
    for pod, cpu_w, mem_w in GL.RESOURCE_LIMITS:  # resourcestats_test.go:203-257 (limits at least the requests here)
        cols = flatten_pods(O.build_pod_objects(hdr, res, [pod]))
        assert cols["lim_cpu_milli"][0] == max(cpu_w, cols["req_cpu_milli"][0]) and cols["lim_mem"][0] == max(mem_w, cols["req_mem"][0])


def test_nodes_match_oracle(hdr, oracle):
    n = 500
    nodes = synth.synth_nodes(hdr, n, seed=6)
    node_pods = synth.synth_node_pods(hdr, n, seed=6)
    cols = flatten_nodes(nodes, node_pods)
    out = oracle.header().structs["orc_node_requests_limits"]()
    zero = (C.c_int64 * 4)(0, 0, 0, 0)
    for i in range(n):
        oracle.lib().orc_node_requests_and_limits(nodes.ref(), node_pods.ref(), i, zero, C.byref(out))
        # with a zero pending pod NodeLimit is the node's own sum; NodeRequest the sum capped by the allocatable
        assert (out.lim_cpu, out.lim_mem) == (cols["lim_cpu_milli"][i], cols["lim_mem"][i])
        assert out.req_cpu == min(cols["req_cpu_milli"][i], nodes.struct.alloc_cpu_milli[i])
        assert out.req_mem == min(cols["req_mem"][i], nodes.struct.alloc_mem[i])
    assert (cols["lim_cpu_milli"] > np.ctypeslib.as_array(nodes.struct.alloc_cpu_milli, (n,))).any()  # over-committed nodes exist


def test_reference_node_fixtures(hdr):
    res = O.Resources()
    for case in GL.NODE_REQUESTS_LIMITS:  # resourcestats_test.go:374-603: the *MinusPod fields are the node's own sums
        nodes = O.build_node_objects(hdr, res, [O.node(case["node"])])
        cols = flatten_nodes(nodes, O.build_node_pods_objects(hdr, res, 1, {0: case["on_node"]}))
        w = case["want"]
        assert min(cols["req_cpu_milli"][0], w["cap_cpu"]) == w["req_minus_pod_cpu"] and min(cols["req_mem"][0], w["cap_mem"]) == w["req_minus_pod_mem"]
        assert cols["lim_cpu_milli"][0] == w["lim_minus_pod_cpu"] and cols["lim_mem"][0] == w["lim_minus_pod_mem"]


def test_no_pods_on_nodes(hdr):
    nodes = synth.synth_nodes(hdr, 10, seed=1)
    cols = flatten_nodes(nodes, None)
    assert not any(v.any() for v in cols.values())
