"""rocprofv3 target: the sequential commit loop of the full profile on config #5's node count, a bounded number of pods.
usage: python tools/prof_commit.py [n_pods]  (run under `rocprofv3 --kernel-trace --stats`)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import scheduler_plugins_amd as spx
from scheduler_plugins_amd import synth, objects as O
from scheduler_plugins_amd.engine import Engine, mask_of

n_pods = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
hdr = spx.header()
snap = synth.full_snapshot(hdr, 20_000, n_pods, quota_sized_for_batch=True)
params = O.nrt_params(hdr, O.Resources(), "LeastAllocated")
with Engine(0) as e:
    if len(sys.argv) > 2 and sys.argv[2] == "direct":  # plain launches instead of the replayed graph (rocprofv3 7.2 crashes on the capture)
        e.set_option("COMMIT_FROM_MEMORY", 1)
    e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
    e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], params)
    e.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
    e.load_quota_objects(snap["pods"], snap["rc"], snap["quota"])
    mask = mask_of(0, 1, 2, 3, 4, 5)
    t = time.perf_counter()
    node, score, ties, _ = e.commit_sequential(mask)
    dt = time.perf_counter() - t
    print(f"{n_pods} pods in {dt*1e3:.1f} ms = {dt/n_pods*1e6:.1f} us/pod; unschedulable {(node < 0).sum()}")
