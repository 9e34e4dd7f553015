#!/usr/bin/env python3
"""round 3 experiment: where does the NRT sweep's time go?  config #3's node snapshot against pod batches of one QoS class each
(BestEffort = the pod loop's fixed overhead, Burstable = item decode without zone arithmetic, Guaranteed = all of it)."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import os
import scheduler_plugins_amd as spx
if os.environ.get('SPX_VARIANT'):
    spx.LIB_PATH = Path(__file__).resolve().parent / '_var' / f"libspx_{os.environ['SPX_VARIANT']}.so"
from scheduler_plugins_amd import synth, objects as O
from scheduler_plugins_amd.engine import Engine, NRT, mask_of

hdr = spx.header()
N, P = 5000, 50000
strategies = sys.argv[1:] or ["LeastAllocated"]
snap = synth.nrt_snapshot(hdr, N, P, seed=synth.SEED)
out = {}
with Engine(0) as e:
    if os.environ.get('SPX_NOSIDE'):
        e.set_option('NRT_SIDE_STREAM', 0)
    if os.environ.get('SPX_SINGLE'):
        e.set_option('NRT_SINGLE_LAUNCH', 1)
    for strat in strategies:
        for name, qp in ((("mix", (0.5, 0.4, 0.1)),) if os.environ.get("SPX_QOS_ONLY") else (("mix", (0.5, 0.4, 0.1)), ("guaranteed", (1, 0, 0)), ("burstable", (0, 1, 0)), ("besteffort", (0, 0, 1)))):
            pods = synth.synth_pods(hdr, P, seed=synth.SEED, device_res=synth.RES_DEVICE, hugepage_res=synth.RES_HUGEPAGES_2MI, qos_p=qp)
            e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], pods, O.nrt_params(hdr, O.Resources(), strat))
            for _ in range(3):
                e.eval(mask_of(NRT))
            e.sync()
            ts = []
            for _ in range(10):
                e.eval(mask_of(NRT))
                e.sync()
                ts.append(e.last_eval_ms())
            out[f"{strat}/{name}"] = round(float(np.median(ts)), 4)
            print(strat, name, out[f"{strat}/{name}"], flush=True)
print(json.dumps(out))
