"""config #2's decisions-only sweep (spx_decide, Allocatable + TLP) in a loop: for rocprofv3 passes.  usage: decide_loop.py [n]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import scheduler_plugins_amd as spx
from scheduler_plugins_amd import synth
from scheduler_plugins_amd.engine import ALLOCATABLE, TLP, Engine, mask_of
hdr = spx.header()
snap = synth.trimaran_snapshot(hdr, 10_000, 100_000)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
with Engine(0) as e:
    e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
    m = mask_of(ALLOCATABLE, TLP)
    for _ in range(3):
        e.decide(m)
    e.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        e.decide(m)
    e.sync()
    print("decide_ms", (time.perf_counter() - t0) * 1e3 / n)
