#!/usr/bin/env python3
"""host side of the full profile at config #5's node count, 8 192 pods: spx_load_* one after the other vs spx_load_profile"""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench
import scheduler_plugins_amd as spx
from scheduler_plugins_amd.engine import Engine
hdr = spx.header()
w = dict(bench.WORKLOADS["config5_share"], n_pods=8192)
snap = bench.build_snapshot(hdr, w, 8192, bench.synth_seed())
with Engine(0) as e:
    out = {}
    for name, conc in (("load_c_ms", False), ("load_profile_ms", True)):
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            e.load_c(snap, snap["nrt_params"], concurrent=conc)
            e.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        out[name] = sorted(ts)[3]
        out[name + "_all"] = [round(x, 2) for x in ts]
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        e._ck(e._lib.spx_load_nrt(e._h, snap["nodes"].ref(), snap["nrt"].ref(), snap["rc"].ref(), snap["pods"].ref(), snap["nrt_params"].ref()))
        e.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    out["load_nrt_ms"] = sorted(ts)[2]
    out["nrt_stages"] = e.last_load_nrt_ms()
print(json.dumps(out))
