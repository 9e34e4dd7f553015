#!/usr/bin/env python3
"""config #5's share (20 000 nodes x 62 500 pods): the per-row kernels of the full profile by difference of evaluations —
NRT alone, + NetworkOverhead (k_net_cls), + Allocatable (k_alloc_masked), + the trimaran pair — from spx_last_eval_ms (HIP events).
Run through tools/variant.py to time a diagnostic build: `python tools/variant.py run <name> tools/r5/time_row_kernels.py`."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench
import scheduler_plugins_amd as spx
from scheduler_plugins_amd.engine import ALLOCATABLE, LVRB, NETOVERHEAD, NRT, TLP, Engine, mask_of

hdr = spx.header()
w = bench.WORKLOADS["config5_share"]
snap = bench.build_snapshot(hdr, w, w["n_pods"], bench.synth_seed())
with Engine(0) as e:
    bench.load_tables(e, w, snap)

    def ms(mask, reps=6):
        best = []
        for _ in range(reps):
            e.eval(mask)
            e.sync()
            best.append(e.last_eval_ms())
        return sorted(best)[len(best) // 2]

    t_nrt = ms(mask_of(NRT))
    t_net = ms(mask_of(NRT, NETOVERHEAD))
    t_all = ms(mask_of(NRT, NETOVERHEAD, ALLOCATABLE))
    t_full = ms(mask_of(NRT, NETOVERHEAD, ALLOCATABLE, TLP, LVRB))
    import time
    full = 0
    for p_ in w["plugins"]:
        full |= 1 << bench.PID[p_]
    for reps in (5, 100):
        e.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            e.eval(full)
        e.sync()
        print(f"bench mask x{reps}: {(time.perf_counter() - t0) * 1e3 / reps:.3f} ms per eval; last_eval_ms {e.last_eval_ms():.3f}")
    print("after the sustained run: nrt+net+alloc", round(ms(mask_of(NRT, NETOVERHEAD, ALLOCATABLE)), 3), " nrt", round(ms(mask_of(NRT)), 3))
    print(f"nrt {t_nrt:.3f}  +net {t_net - t_nrt:.3f}  +alloc_masked {t_all - t_net:.3f}  +tlp,lvrb {t_full - t_all:.3f}  total {t_full:.3f} ms")
