"""TopologyMatch.Filter inside a preemption dry-run, the reference's TestFilter_PreemptionFlow as data
(tests/golden/nrt_preemption_flow.py): the dispatch of filter.go:205-220 restated around the product's eviction simulation
(spx_nrt_post_eviction, host/nrt_preemption.cc) and the oracle's Filter on the zone table it returns.  CPU only — on the GPU the
second half is the ordinary NRT Filter sweep on those availabilities (tests/test_gpu_nrt.py::test_preemption_dry_run_filter)."""
import ctypes as C

import numpy as np
import pytest

import scheduler_plugins_amd as spx
from golden import nrt_preemption_flow as GF
from golden.nrt_preemption import ERROR_CODES
from helpers import NRT
from scheduler_plugins_amd import objects as O
from test_oracle_golden_nrt import MSG

UNKNOWN = -2  # SPX_EVICT_CTR_UNKNOWN
MESSAGE_OF = {code: msg for msg, code in ERROR_CODES.items()}


def filter_status(hdr, oracle, nrt_table, res, preemptor):
    pods = O.build_pod_objects(hdr, res, [{"containers": [O.container(c["requests"], c["limits"]) for c in preemptor["containers"]]}])
    node = O.build_node_objects(hdr, res, [O.node(GF.NODE)])
    snap = oracle.Snapshot(node, pods, rc=res.table(hdr), nrt=nrt_table, nrt_params=O.nrt_params(hdr, res, "LeastAllocated"))
    return int(snap.filter_rows(NRT)[0, 0])


@pytest.mark.parametrize("case", GF.CASES, ids=lambda c: f"L{c['line']}")
def test_filter_in_the_preemption_flow(hdr, oracle, case):
    res = O.Resources()
    live = O.build_nrt_objects(hdr, res, [O.nrt(GF.NRT["zones"], GF.NRT["policies"])])
    # getVictimPods: the stack is only read when preemption is enabled (prefilter.go / filter.go:205-209)
    victims = case["victims"] if case["enabled"] else []
    table, message = live, None
    placement = case["placement"]
    if victims and placement is not None and len(placement) != 0:      # filter.go:210-212
        vt = O.build_pod_objects(hdr, res, [{"containers": [O.container(c["requests"], c["limits"]) for c in v["containers"]]} for v in victims])
        qos = np.zeros(len(victims), np.uint8)                          # makeGuaranteedPod
        numa = np.array([placement.get((v["ns"], v["name"], c["name"]), UNKNOWN) for v in victims for c in v["containers"]] + [0], np.int32)
        n_entries = 4                                                   # two zones x (cpu, memory)
        out = np.zeros(n_entries, np.int64)
        code = C.c_int32(-1)
        assert spx.lib().spx_nrt_post_eviction(live.ref(), res.table(hdr).ref(), 0, vt.ref(), qos.ctypes.data_as(C.POINTER(C.c_uint8)),
                                               numa.ctypes.data_as(C.POINTER(C.c_int32)), 1, len(placement),
                                               out.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(code)) == 0
        if code.value != 0:
            message = "eviction simulation in NRT is not possible:" + MESSAGE_OF[code.value]      # filter.go:214-216
        else:
            avail = np.ctypeslib.as_array(live.struct.zres_avail, (n_entries,))
            assert out.tolist() != avail.tolist()                       # the simulation gave something back
            qty = lambda name, v: f"{int(v)}m" if name == "cpu" else str(int(v))     # the table holds cpu in millicores
            zones = [dict(z, resources=[(n, cap, alloc, qty(n, out[2 * zi + ri])) for ri, (n, cap, alloc, _) in enumerate(z["resources"])])
                     for zi, z in enumerate(GF.NRT["zones"])]
            table = O.build_nrt_objects(hdr, res, [O.nrt(zones, GF.NRT["policies"])])
            assert np.ctypeslib.as_array(table.struct.zres_avail, (n_entries,)).tolist() == out.tolist()
    if message is None:
        st = filter_status(hdr, oracle, table, res, case["preemptor"])
        message = None if st == 0 else next(m for m, c in MSG.items() if c == st)
    assert message == case["want"]
    # NodeMaybeOverReserved is called for a failed Filter outside the preemption flow only (filter.go:241-243): the Go side's
    # business; what the fixture pins is which cases ARE a preemption flow
    if case["over_reserved"] is not None:
        assert (message is not None and not victims) == case["over_reserved"]
