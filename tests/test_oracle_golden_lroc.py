"""Pins the LowRiskOverCommitment oracle (oracle/orc_lroc.c) against the reference's own tables and, for the third-party
incomplete beta function, against an independent implementation (tests/golden/gen_betainc.py)."""
import ctypes as C
import json
import math
from pathlib import Path

import numpy as np
import pytest

from golden import lroc as GL
from helpers import LROC, lroc_params
from scheduler_plugins_amd import objects as O

MT = {"CPU": 0, "Memory": 1}


def _f(lib, name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype, f.argtypes = restype, list(argtypes)
    return f


@pytest.mark.parametrize("case", GL.MATCH_MOMENTS, ids=lambda c: c[0])
def test_match_moments(oracle, case):
    _, m1, m2, want, alpha_w, beta_w = case
    a, b = C.c_double(), C.c_double()
    got = oracle.lib().orc_beta_match_moments(C.c_double(m1), C.c_double(m2), C.byref(a), C.byref(b))
    assert bool(got) == want
    if want:
        assert abs(a.value - alpha_w) < 1e-12 and abs(b.value - beta_w) < 1e-12


@pytest.mark.parametrize("case", GL.DISTRIBUTION_FUNCTION, ids=lambda c: c[0])
def test_distribution_function(oracle, case):
    _, alpha, beta, x, want = case
    f = _f(oracle.lib(), "orc_beta_distribution_function", C.c_double, C.c_double, C.c_double, C.c_double)
    assert abs(f(alpha, beta, x) - want) <= GL.TOLERANCE


@pytest.mark.parametrize("m1,want", GL.MAX_VARIANCE)
def test_max_variance(oracle, m1, want):
    f = _f(oracle.lib(), "orc_beta_max_variance", C.c_double, C.c_double)
    assert abs(f(m1) - want) <= GL.TOLERANCE


def test_reg_inc_beta_against_independent_implementation(oracle):
    f = _f(oracle.lib(), "orc_reg_inc_beta", C.c_double, C.c_double, C.c_double, C.c_double)
    pts = json.loads((Path(__file__).parent / "golden" / "betainc.json").read_text())
    assert len(pts) == 600
    worst = max(abs(f(p["a"], p["b"], p["x"]) - p["value"]) for p in pts)
    assert worst < 2e-9, worst
    # symmetry I_x(a,b) = 1 - I_{1-x}(b,a) and monotonicity in x, at sizes the fixture does not reach
    for a, b in [(0.3, 7.0), (2500.0, 900.0), (5.0, 5.0)]:
        xs = np.linspace(0.01, 0.99, 99)
        v = np.array([f(a, b, x) for x in xs])
        assert (np.diff(v) >= -1e-12).all()
        w = np.array([1 - f(b, a, 1 - x) for x in xs])
        assert np.abs(v - w).max() < 1e-9


@pytest.mark.parametrize("case", GL.SCORE_CASES, ids=lambda c: f"L{c['line']}")
def test_score(hdr, oracle, case):
    res = O.Resources()
    nodes = O.build_node_objects(hdr, res, [O.node(case["node"])])
    pods = O.build_pod_objects(hdr, res, [case["pod"]])
    snap = oracle.Snapshot(nodes, pods, metrics=O.build_metrics_objects(hdr, 1, case["metrics"]),
                           node_pods=O.build_node_pods_objects(hdr, res, 1, {}), lroc_params=lroc_params(hdr))
    raw, norm = snap.score_rows(LROC)
    assert raw[0].tolist() == case["expected"] and norm[0].tolist() == case["expected"]


@pytest.mark.parametrize("case", GL.COMPUTE_RISK, ids=lambda c: c[0])
def test_compute_risk(hdr, oracle, case):
    _, mtype, nrla, want = case
    res = O.Resources()
    nodes = O.build_node_objects(hdr, res, [O.node(GL.NODE_A)])
    metrics = O.build_metrics_objects(hdr, 1, {0: GL.METRICS_A})
    nrl = oracle.header().structs["orc_node_requests_limits"](**nrla)
    f = oracle.lib().orc_lroc_compute_risk
    f.restype = C.c_double
    got = f(nodes.ref(), metrics.ref(), 0, MT[mtype], C.byref(nrl), lroc_params(hdr).ref())
    assert got == want  # the reference compares with ==


@pytest.mark.parametrize("i", range(len(GL.RESOURCE_LIMITS)))
def test_get_resource_limits(hdr, oracle, i):
    pod, cpu_w, mem_w = GL.RESOURCE_LIMITS[i]
    res = O.Resources()
    pods = O.build_pod_objects(hdr, res, [pod])
    cpu, mem = C.c_int64(), C.c_int64()
    oracle.lib().orc_get_resource_limits(pods.ref(), 0, C.byref(cpu), C.byref(mem))
    assert (cpu.value, mem.value) == (cpu_w, mem_w)


def pod_rl(oracle, pods, i=0):
    """CreatePodResourcesStateData: requests, and limits raised to them"""
    v = [C.c_int64() for _ in range(4)]
    oracle.lib().orc_get_resource_requested(pods.ref(), i, C.byref(v[0]), C.byref(v[1]))
    oracle.lib().orc_get_resource_limits(pods.ref(), i, C.byref(v[2]), C.byref(v[3]))
    r = [x.value for x in v]
    return [r[0], r[1], max(r[2], r[0]), max(r[3], r[1])]


@pytest.mark.parametrize("case", GL.NODE_REQUESTS_LIMITS, ids=lambda c: c["name"])
def test_node_requests_and_limits(hdr, oracle, case):
    res = O.Resources()
    nodes = O.build_node_objects(hdr, res, [O.node(case["node"])])
    pods = O.build_pod_objects(hdr, res, [case["pod"]])
    node_pods = O.build_node_pods_objects(hdr, res, 1, {0: case["on_node"]})
    rl = (C.c_int64 * 4)(*pod_rl(oracle, pods))
    out = oracle.header().structs["orc_node_requests_limits"]()
    oracle.lib().orc_node_requests_and_limits(nodes.ref(), node_pods.ref(), 0, rl, C.byref(out))
    assert {k: getattr(out, k) for k in case["want"]} == case["want"]


@pytest.mark.parametrize("case", GL.COMPUTE_RISK_AS_PODS, ids=lambda c: c["name"])
def test_compute_risk_fixtures_through_score(hdr, oracle, case):
    res = O.Resources()
    nodes = O.build_node_objects(hdr, res, [O.node(GL.NODE_A)])
    pods = O.build_pod_objects(hdr, res, [case["pod"]])
    node_pods = O.build_node_pods_objects(hdr, res, 1, {0: case["on_node"]})
    snap = oracle.Snapshot(nodes, pods, metrics=O.build_metrics_objects(hdr, 1, {0: GL.METRICS_A}), node_pods=node_pods,
                           lroc_params=lroc_params(hdr))
    assert snap.score_rows(LROC)[0][0].tolist() == [case["score"]]
    # and the intermediate sums are the fixture's
    rl = (C.c_int64 * 4)(*pod_rl(oracle, pods))
    out = oracle.header().structs["orc_node_requests_limits"]()
    oracle.lib().orc_node_requests_and_limits(nodes.ref(), node_pods.ref(), 0, rl, C.byref(out))
    want = GL.NRLA_A1 if case["name"] == "nrla_A1" else GL.NRLA_A2
    assert {k: getattr(out, k) for k in want} == want
