#!/usr/bin/env python3
"""how many cells the LeastNUMANodes batch Score launch lists for k_nrt_ln_redo (spx_fetch_stats), config #3's shape"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import scheduler_plugins_amd as spx
from scheduler_plugins_amd import synth, objects as O
from scheduler_plugins_amd.engine import Engine, mask_of
hdr = spx.header()
N, P = 5000, 50000
snap = synth.nrt_snapshot(hdr, N, P, seed=synth.SEED)
with Engine(0) as e:
    e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], O.nrt_params(hdr, O.Resources(), "LeastNUMANodes"))
    e.stats(reset=True)
    e.eval(mask_of(3)); e.sync()
    st = e.stats(reset=True)
    u, d = e.nrt_pod_classes()
    print("listed cells", int(st[3]), "rows evaluated", u, "cells evaluated", u * N, "share", st[3] / (u * N))
