#!/usr/bin/env python3
"""round 3 experiment: the NRT sweep on a batch of BestEffort pods only (nothing to compute), per library variant"""
import os, sys, json
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
snap = synth.nrt_snapshot(hdr, N, P, seed=synth.SEED)
with Engine(0) as e:
    for name, qp in (("besteffort", (0, 0, 1)), ("mix", (0.5, 0.4, 0.1))):
        pods = synth.synth_pods(hdr, P, seed=synth.SEED, device_res=synth.RES_DEVICE, hugepage_res=synth.RES_HUGEPAGES_2MI, qos_p=qp)
        e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], pods, O.nrt_params(hdr, O.Resources(), "LeastAllocated"))
        for _ in range(3):
            e.eval(mask_of(NRT))
        e.sync()
        ts = []
        for _ in range(10):
            e.eval(mask_of(NRT)); e.sync(); ts.append(e.last_eval_ms())
        print(os.environ.get('SPX_VARIANT', ''), name, round(float(np.median(ts)), 4), flush=True)
