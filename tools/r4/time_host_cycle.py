"""Host side of a config #2 cycle, piece by piece (median of 30): the node delta and the new pending batch.
   python tools/r4/time_host_cycle.py [n_nodes n_pods]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scheduler_plugins_amd import synth  # noqa: E402
import scheduler_plugins_amd as spx  # noqa: E402
from scheduler_plugins_amd.engine import Engine  # noqa: E402

n_nodes, n_pods = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 100000)
hdr = spx.header()
snap = synth.trimaran_snapshot(hdr, n_nodes, n_pods)
pods2 = synth.synth_pods(hdr, n_pods, seed=synth.SEED + 17)
idx = np.sort(np.random.default_rng(7).choice(n_nodes, max(1, n_nodes // 100), replace=False))


def med(f, n=30):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 4)


with Engine(0) as e:
    e.load_trimaran_objects(snap["nodes"], snap.get("rc"), snap["pods"], snap["metrics"], snap.get("assigned"))
    e.sync()
    rows = e.flatten_trimaran_node_rows(snap["nodes"], snap["metrics"], snap.get("assigned"), idx)
    cols = e.flatten_trimaran_pods(pods2)
    out = {
        "flatten_node_rows_ms": med(lambda: e.flatten_trimaran_node_rows(snap["nodes"], snap["metrics"], snap.get("assigned"), idx)),
        "update_node_rows_ms": med(lambda: e.update_trimaran_node_rows(idx, rows)),
        "flatten_pods_ms": med(lambda: e.flatten_trimaran_pods(pods2)),
        "upload_pods_ms": med(lambda: e.upload_trimaran_pods(cols)),
        "load_pods_ms": med(lambda: e.load_trimaran_pods(pods2)),
    }
    print(out)
