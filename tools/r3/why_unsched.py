#!/usr/bin/env python3
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import scheduler_plugins_amd as spx
from scheduler_plugins_amd import synth, objects as O
from scheduler_plugins_amd.engine import Engine, mask_of, ALLOCATABLE, TLP, LVRB, NRT, NETOVERHEAD, CAPACITY
hdr = spx.header()
N, P = 20000, 62500
for sized in (True,):
    snap = synth.full_snapshot(hdr, N, P, seed=synth.SEED, quota_sized_for_batch=sized)
    params = O.nrt_params(hdr, O.Resources(), "LeastAllocated")
    with Engine(0) as e:
        e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], params)
        e.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
        e.load_quota_objects(snap["pods"], snap["rc"], snap["quota"])
        mask = mask_of(ALLOCATABLE, TLP, LVRB, NRT, NETOVERHEAD, CAPACITY)
        e.eval(mask); e.eval_best(mask); e.sync()
        node, score, ties, feas = e.best()
        pf = e.prefilter(CAPACITY)
        print("sized", sized, "frozen: prefilter rejected", int((pf != 0).sum()), "no feasible node", int((node < 0).sum()), "median feasible", int(np.median(feas)))
        st_n = e.all_status(NRT, 0, 2000); st_w = e.all_status(NETOVERHEAD, 0, 2000)
        print("   NRT infeasible share", float((st_n != 0).mean()), "Net infeasible share", float((st_w != 0).mean()))
        seq = e.commit_sequential(mask, 0, P, want_ties=False)[0]
        un = seq < 0
        print("   sequential all: unschedulable", int(un.sum()), "by 1000s", [int(un[i:i+1000].sum()) for i in range(0, P, 6250)])
