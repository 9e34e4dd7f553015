import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import scheduler_plugins_amd as spx
from scheduler_plugins_amd import synth, objects as O
from scheduler_plugins_amd.engine import Engine, mask_of
NRT = 3
hdr = spx.header()
n_nodes, n_pods = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1500, 2500)
snap = synth.nrt_snapshot(hdr, n_nodes, n_pods, seed=37)
params = O.nrt_params(hdr, O.Resources(), "BalancedAllocation")
with Engine(0) as e:
    e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], params)
    for fused in (1, 0, 1):
        e.set_option("NRT_FUSED", fused)
        e.stats(reset=True)
        e.eval(mask_of(NRT)); e.sync()
        t0 = time.time()
        for _ in range(3):
            e.eval(mask_of(NRT))
        e.sync()
        dt = (time.time() - t0) / 3 * 1e3
        sc = e.all_scores(NRT)
        print("fused", fused, "path", e.nrt_filter_path(), "ms", round(dt, 3), "stats", e.stats()[NRT], "cells", sc.size, "255s", int((sc == 255).sum()), "mean", float(sc.mean()))
