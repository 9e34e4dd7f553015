"""spx_commit_sequential of the full profile on config #5's node count: the cooperative persistent kernel against the per-pod launches
replayed from a graph, every decision compared.  usage: python tools/r4/time_commit_coop.py [n_pods] [n_nodes] [graph_pods]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import scheduler_plugins_amd as spx
from scheduler_plugins_amd import synth, objects as O
from scheduler_plugins_amd.engine import Engine, mask_of

n_pods = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
n_nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
graph_pods = int(sys.argv[3]) if len(sys.argv) > 3 else min(n_pods, 4000)
hdr = spx.header()
snap = synth.full_snapshot(hdr, n_nodes, n_pods, quota_sized_for_batch=True)
params = O.nrt_params(hdr, O.Resources(), "LeastAllocated")
with Engine(0) as e:
    e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
    e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], params)
    e.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
    e.load_quota_objects(snap["pods"], snap["rc"], snap["quota"])
    mask = mask_of(0, 1, 2, 3, 4, 5)
    out = {}
    for name, opt, rows in (("coop", 1, n_pods), ("coop", 1, n_pods), ("graph", 0, graph_pods)):
        e.set_option("COMMIT_COOP", opt)
        t = time.perf_counter()
        node, score, ties, missing = e.commit_sequential(mask, 0, rows)
        dt = time.perf_counter() - t
        out[name] = (node, score, ties, missing)
        print(f"{name}: path {e.commit_path()}: {n_nodes} nodes x {rows} pods in {dt*1e3:.1f} ms = {dt/rows*1e6:.2f} us/pod; unschedulable {(node < 0).sum()}; "
              f"distinct nodes {len(set(node.tolist()))}", flush=True)
    a, b = out["coop"], out["graph"]
    g = graph_pods
    same = np.array_equal(a[0][:g], b[0]) and np.array_equal(a[1][:g], b[1]) and np.array_equal(a[2][:g], b[2])
    print("coop == graph on the first", g, "pods:", same)
    if not same:
        bad = np.flatnonzero((a[0][:g] != b[0]) | (a[1][:g] != b[1]) | (a[2][:g] != b[2]))
        print("first mismatches:", bad[:10], a[0][bad[:5]], b[0][bad[:5]], a[1][bad[:5]], b[1][bad[:5]], a[2][bad[:5]], b[2][bad[:5]])
        sys.exit(1)
    if g == n_pods:
        print("missing equal:", np.array_equal(a[3], b[3]))
