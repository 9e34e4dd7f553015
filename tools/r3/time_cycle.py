#!/usr/bin/env python3
"""round 3: where config5_share's 88 ms of flatten + upload go — every host flattener and every upload call timed separately"""
import sys, time, json
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
T = {}
def timed(name, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t0)
    T[name] = round(best * 1e3, 2)
    return r
with Engine(0) as e:
    e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])  # shape
    a = timed("flatten_alloc_nodes", lambda: e.flatten_alloc_nodes(snap["nodes"], snap["rc"]))
    tn = timed("flatten_trimaran_nodes", lambda: e.flatten_trimaran_nodes(snap["nodes"], snap["metrics"], snap["assigned"]))
    tp = timed("flatten_trimaran_pods", lambda: e.flatten_trimaran_pods(snap["pods"]))
    timed("upload_alloc_nodes", lambda: (e.upload_alloc_nodes(a), e.sync()))
    timed("upload_trimaran_nodes", lambda: (e.upload_trimaran_nodes(tn), e.sync()))
    timed("upload_trimaran_pods", lambda: (e.upload_trimaran_pods(tp), e.sync()))
    f = timed("flatten_nrt(slots+nodes+pods)", lambda: e.flatten_nrt(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], params))
    L, H = e._lib, e._hdr
    from scheduler_plugins_amd._abi import Table
    e._ck(L.spx_set_nrt_params(e._h, f["params"].ref())); e._ck(L.spx_upload_nrt_slots(e._h, f["slots"].ref()))
    tnodes = Table(H, "spx_nrt_nodes_soa", n_nodes=f["N"], n_res=f["R"], **f["nodes"])
    tpods = Table(H, "spx_nrt_pods_soa", n_pods=P, n_res=f["R"], **f["pods"])
    timed("upload_nrt_nodes", lambda: (e._ck(L.spx_upload_nrt_nodes(e._h, tnodes.ref())), e.sync()))
    timed("upload_nrt_pods", lambda: (e._ck(L.spx_upload_nrt_pods(e._h, tpods.ref())), e.sync()))
    g = timed("flatten_network", lambda: e.flatten_network(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"]))
    timed("upload_network", lambda: (e.upload_network(g), e.sync()))
    q = timed("flatten_quota", lambda: e.flatten_quota(snap["pods"], snap["rc"], snap["quota"]))
    timed("upload_quota", lambda: (e.upload_quota(q), e.sync()))
print(json.dumps(T, indent=1)); print("total", round(sum(T.values()), 1))
