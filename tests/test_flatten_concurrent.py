"""The host flatteners under concurrent callers (the ranks of a MultiEngine load their own pod rows from their own threads; cgo
callers arrive on arbitrary OS threads): the parked worker pools (host/parallel.hpp) serve one job each, further callers run inline —
whichever way a call is served, its columns must be the ones a lone call writes."""
import threading

import numpy as np
import pytest

from helpers import tlp_params
from scheduler_plugins_amd import objects as O
from scheduler_plugins_amd import synth
from test_flatten_nrt_rows import HostOnly


@pytest.mark.parametrize("where", ["here", pytest.param("many_core_box", marks=pytest.mark.gpu)])
def test_concurrent_flatteners_write_what_a_lone_call_writes(hdr, where):
    """(the second variant needs no device: it is marked `gpu` so that it also runs on the GPU box, whose 256 hardware threads give
    the process eight pools — six concurrent callers are then all served by workers; on a small host there is one pool)"""
    n_nodes, n_pods = 400, 70_000  # enough rows for the flatteners to go parallel (8192 rows per thread)
    snap = synth.nrt_snapshot(hdr, n_nodes, n_pods, seed=21)
    params = O.nrt_params(hdr, O.Resources(), "LeastAllocated")

    def engine():
        e = HostOnly()
        e.tlp_params = tlp_params(hdr, 40, 1000, 1.5)
        return e

    e0 = engine()
    want_tri = e0.flatten_trimaran_pods(snap["pods"])
    want_nrt = e0.flatten_nrt(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], params)["pods"]
    views = [(0, 20_000), (20_000, 45_000), (45_000, 70_000)]
    results, errors = {}, []

    def work(k):
        try:
            e = engine()
            for rep in range(3):
                if k < 3:   # whole-table calls racing each other
                    results[(k, rep)] = ("tri", None, e.flatten_trimaran_pods(snap["pods"]))
                else:       # a rank's view of the batch
                    b, en = views[k - 3]
                    view = O.pod_rows(hdr, snap["pods"], b, en)
                    results[(k, rep)] = ("nrt", (b, en), e.flatten_nrt(snap["nodes"], snap["nrt"], snap["rc"], view, params))
        except BaseException as ex:
            errors.append(ex)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(results) == 18
    for kind, rows, got in results.values():
        if kind == "tri":
            for c in want_tri:
                assert np.array_equal(got[c], want_tri[c]), c
        else:
            b, en = rows
            assert got["P"] == en - b
            if got["R"] == len(want_nrt["pod_req"]) // n_pods:
                for c, v in want_nrt.items():
                    per = len(v) // n_pods
                    assert np.array_equal(got["pods"][c], v.reshape(n_pods, per)[b:en].reshape(-1)), c


def test_flatteners_after_fork(hdr):
    """a child forked after the parent used the worker pools (whose threads do not exist in the child) starts pools of its own
    (pthread_atfork in host/parallel.hpp): the flattener returns, and returns the lone call's columns — instead of waiting forever on
    workers that are not there"""
    import os
    import signal
    snap = synth.trimaran_snapshot(hdr, 50, 70_000, seed=5)
    e = HostOnly()
    e.tlp_params = tlp_params(hdr, 40, 1000, 1.5)
    want = e.flatten_trimaran_pods(snap["pods"])   # parent: pools created, workers parked
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:  # child
        ok = b"0"
        try:
            signal.alarm(60)   # a deadlock ends the child instead of hanging the suite
            got = e.flatten_trimaran_pods(snap["pods"])
            ok = b"1" if all(np.array_equal(got[c], want[c]) for c in want) else b"2"
        finally:
            os.write(w, ok)
            os._exit(0)
    os.close(w)
    _, status = os.waitpid(pid, 0)
    assert os.read(r, 1) == b"1" and status == 0
    again = e.flatten_trimaran_pods(snap["pods"])   # and the parent's pools still work
    assert all(np.array_equal(again[c], want[c]) for c in want)
