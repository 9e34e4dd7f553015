"""Per-call wall time of the library's one-call loaders (spx_load_trimaran / _nrt / _network / _quota: what a cgo caller pays for a full
snapshot) at config #5's node count.  usage: python tools/r5/time_load_c.py [n_nodes] [n_pods]   (on the GPU box)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import scheduler_plugins_amd as spx
from scheduler_plugins_amd import synth, objects as O
from scheduler_plugins_amd.engine import Engine

n_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
n_pods = int(sys.argv[2]) if len(sys.argv) > 2 else 8_192
hdr = spx.header()
snap = synth.full_snapshot(hdr, n_nodes, n_pods, quota_sized_for_batch=True)
params = O.nrt_params(hdr, O.Resources(), "LeastAllocated")
ref = lambda t: t.ref() if t is not None else None
with Engine(0) as e:
    L, h = e._lib, e._h
    calls = {
        "trimaran": lambda: L.spx_load_trimaran(h, snap["nodes"].ref(), ref(snap.get("rc")), snap["pods"].ref(), snap["metrics"].ref(), ref(snap.get("assigned"))),
        "nrt": lambda: L.spx_load_nrt(h, snap["nodes"].ref(), snap["nrt"].ref(), ref(snap.get("rc")), snap["pods"].ref(), params.ref()),
        "network": lambda: L.spx_load_network(h, snap["nodes"].ref(), snap["pods"].ref(), snap["appgroups"].ref(), snap["nettopo"].ref()),
        "quota": lambda: L.spx_load_quota(h, snap["pods"].ref(), ref(snap.get("rc")), snap["quota"].ref()),
    }
    out = {}
    for rep in range(4):
        for k, fn in calls.items():
            e.sync()
            t0 = time.perf_counter()
            assert fn() == 0
            e.sync()
            out.setdefault(k, []).append((time.perf_counter() - t0) * 1e3)
    import ctypes as C
    ms = (C.c_double * 6)()
    L.spx_last_load_nrt_ms(h, ms)
    print("spx_load_nrt stages (last call): slots %.2f | flatten nodes %.2f | flatten pods %.2f | params+slot table %.2f | upload nodes %.2f | upload pods %.2f ms" % tuple(ms))
    for k, v in out.items():
        print(f"{k:9s} first {v[0]:7.2f} ms   then {sorted(v[1:])[1]:6.2f} ms (median of 3)")
    print("total (median)", round(sum(sorted(v[1:])[1] for v in out.values()), 2), "ms")
