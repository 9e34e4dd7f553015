#!/usr/bin/env python3
"""round 3: the NRT sweep (config #3's nodes) on ONE pod batch, a few launches — the target of rocprofv3 --pmc passes.
usage: exp_one.py <strategy> <g,b,e shares> [launches]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import scheduler_plugins_amd as spx
if os.environ.get('SPX_VARIANT'):
    spx.LIB_PATH = Path(__file__).resolve().parent / '_var' / f"libspx_{os.environ['SPX_VARIANT']}.so"
from scheduler_plugins_amd import synth, objects as O
from scheduler_plugins_amd.engine import Engine, NRT, mask_of
hdr = spx.header()
N, P = 5000, 50000
strat = sys.argv[1]
qp = tuple(float(x) for x in sys.argv[2].split(","))
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 3
snap = synth.nrt_snapshot(hdr, N, P, seed=synth.SEED)
pods = synth.synth_pods(hdr, P, seed=synth.SEED, device_res=synth.RES_DEVICE, hugepage_res=synth.RES_HUGEPAGES_2MI, qos_p=qp)
with Engine(0) as e:
    e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], pods, O.nrt_params(hdr, O.Resources(), strat))
    ts = []
    for _ in range(launches):
        e.eval(mask_of(NRT)); e.sync(); ts.append(e.last_eval_ms())
    print(strat, qp, [round(t, 4) for t in ts])
