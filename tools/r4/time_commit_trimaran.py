"""Wall time of spx_commit_sequential for the Filter-less profile (config #2's shape by default).
usage: python tools/r4/time_commit_trimaran.py [n_nodes] [n_pods] [plugins: e.g. 0,1 or 0,1,2] [ties 0|1] [reps]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import scheduler_plugins_amd as spx
from scheduler_plugins_amd import synth
from scheduler_plugins_amd.engine import Engine, mask_of

n_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
n_pods = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
plugins = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (0, 1)
ties = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
hdr = spx.header()
snap = synth.trimaran_snapshot(hdr, n_nodes, n_pods)
with Engine(0) as e:
    e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
    out = {}
    for state in ("registers", "memory"):
        e.set_option("COMMIT_FROM_MEMORY", 1 if state == "memory" else 0)
        best = 1e9
        for _ in range(reps if state == "registers" else 1):
            t = time.perf_counter()
            node, score, tie, missing = e.commit_sequential(mask_of(*plugins), want_ties=ties)
            best = min(best, time.perf_counter() - t)
        out[state] = (node, score, tie, missing)
        print(f"{state}: {n_nodes} nodes x {n_pods} pods plugins {plugins} ties {ties}: {best*1e3:.1f} ms = {best/n_pods*1e6:.3f} us/pod; distinct nodes {len(set(node.tolist()))}", flush=True)
    a, b = out["registers"], out["memory"]
    same = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]) and (not ties or np.array_equal(a[2], b[2]))
    print("registers == memory:", same)
    if not same:
        bad = np.flatnonzero((a[0] != b[0]) | (a[1] != b[1]))
        print("first mismatches:", bad[:10], a[0][bad[:5]], b[0][bad[:5]], a[1][bad[:5]], b[1][bad[:5]])
        sys.exit(1)
