#!/usr/bin/env python3
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import scheduler_plugins_amd as spx
from scheduler_plugins_amd import synth, objects as O
from scheduler_plugins_amd.engine import Engine
hdr = spx.header()
N, P = 20000, 62500
snap = synth.full_snapshot(hdr, N, P, seed=synth.SEED)
params = O.nrt_params(hdr, O.Resources(), "LeastAllocated")
def T(name, fn, reps=3):
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
        print(f"{name:32s} {dt*1e3:8.2f} ms", flush=True)
    return r
with Engine(0) as e:
    e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
    e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], params)
    idx = np.sort(np.random.default_rng(7).choice(N, N // 100, replace=False))
    cols = T("flatten_trimaran_nodes", lambda: e.flatten_trimaran_nodes(snap["nodes"], snap["metrics"], snap["assigned"]))
    T("update_trimaran_nodes", lambda: e.update_trimaran_nodes(idx, cols))
    slots = e.nrt_soa["slots"]
    rows = T("flatten_nrt_node_rows", lambda: e.flatten_nrt_node_rows(snap["nodes"], snap["nrt"], slots, idx))
    T("update_nrt_node_rows", lambda: e.update_nrt_node_rows(idx, rows, int(slots.struct.n_res)))
