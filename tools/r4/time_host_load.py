"""Where the host side of a full snapshot load goes, per plugin family: flatten (objects -> SoA) and upload (SoA -> device tables,
incl. the engine's own host work: NRT records / classes / node order, LeastNUMANodes tables) — median of 5.
   python tools/r4/time_host_load.py [n_nodes n_pods]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import scheduler_plugins_amd as spx  # noqa: E402
from scheduler_plugins_amd import objects as O, synth  # noqa: E402
from scheduler_plugins_amd.engine import Engine  # noqa: E402

n_nodes, n_pods = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20000, 8192)
hdr = spx.header()
snap = synth.full_snapshot(hdr, n_nodes, n_pods, seed=5, pods_per_group=20, n_namespaces=20)
params = O.nrt_params(hdr, O.Resources(), "LeastAllocated")


def med(f, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 3)


with Engine(0) as e:
    out = {}
    f_nrt = e.flatten_nrt(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], params)
    f_net = e.flatten_network(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
    f_q = e.flatten_quota(snap["pods"], snap["rc"], snap["quota"])
    tri_n = e.flatten_trimaran_nodes(snap["nodes"], snap["metrics"], snap["assigned"])
    tri_p = e.flatten_trimaran_pods(snap["pods"])
    out["trimaran_flatten_nodes"] = med(lambda: e.flatten_trimaran_nodes(snap["nodes"], snap["metrics"], snap["assigned"]))
    out["trimaran_flatten_pods"] = med(lambda: e.flatten_trimaran_pods(snap["pods"]))
    out["trimaran_upload"] = med(lambda: (e.upload_trimaran_nodes(tri_n), e.upload_trimaran_pods(tri_p), e.sync()))
    out["nrt_flatten"] = med(lambda: e.flatten_nrt(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], params))
    out["nrt_upload"] = med(lambda: (e.upload_nrt(f_nrt), e.sync()))
    out["net_flatten"] = med(lambda: e.flatten_network(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"]))
    out["net_upload"] = med(lambda: (e.upload_network(f_net), e.sync()))
    out["quota_flatten"] = med(lambda: e.flatten_quota(snap["pods"], snap["rc"], snap["quota"]))
    out["quota_upload"] = med(lambda: (e.upload_quota(f_q), e.sync()))
    out["one_call_loaders_all"] = med(lambda: (e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"]),
                                               e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], params),
                                               e.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"]),
                                               e.load_quota_objects(snap["pods"], snap["rc"], snap["quota"]), e.sync()))
    print(n_nodes, n_pods, out)
