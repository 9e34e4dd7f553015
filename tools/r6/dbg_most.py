import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import scheduler_plugins_amd as spx
from scheduler_plugins_amd import synth, objects as O
from scheduler_plugins_amd.engine import Engine, mask_of
NRT = 3
hdr = spx.header()
snap = synth.nrt_snapshot(hdr, 333, 160, seed=1)
params = O.nrt_params(hdr, O.Resources(), "MostAllocated")
with Engine(0) as e:
    e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], params)
    e.eval(mask_of(NRT)); e.sync()
    print("path", e.nrt_filter_path(), "pk", e.nrt_packed_score_slots())
    a = e.all_scores(NRT).astype(int); sa = e.all_status(NRT)
    e.set_option("NRT_FUSED", 0)
    e.eval(mask_of(NRT)); e.sync()
    print("path", e.nrt_filter_path())
    b = e.all_scores(NRT).astype(int); sb = e.all_status(NRT)
    bad = np.argwhere(a != b)
    print("status equal", np.array_equal(sa, sb), "bad", len(bad))
    pods = e.nrt_soa["pods"]; nodes = e.nrt_soa["nodes"]
    flags = nodes["flags"]
    rows = sorted(set(bad[:, 0].tolist()))
    print("rows with mismatches", rows[:20], "of", len(rows))
    for p, n in bad[:12]:
        print("pod", p, "node", n, "fused", a[p, n], "two-launch", b[p, n], "n_ctr", pods["n_ctr"][p], "kinds", pods["ctr_kind"][p*8:p*8+4].tolist(), "pod_scope", bool(flags[n] & 8), "status", sa[p, n])
    p = int(bad[0][0])
    R = int(e.nrt_soa["slots"].struct.n_res)
    print("ctr_req", pods["ctr_req"][p*8*R:(p*8+4)*R].reshape(4, R).tolist(), "present", pods["ctr_present"][p*8:p*8+4].tolist())
    print("pod_req", pods["pod_req"][p*R:(p+1)*R].tolist())
    for n in (35, 77):
        nz = int(nodes["n_zones"][n]); Z = 8
        av = nodes["zone_avail"][n*Z*R:(n+1)*Z*R].reshape(Z, R)
        zp = nodes["zone_present"][n*Z:(n+1)*Z]
        print("node", n, "nz", nz, "present", zp.tolist(), "node_present", nodes["node_present"][n])
        print(av.tolist())
        req = pods["pod_req"][p*R:(p+1)*R]
        pres = int(pods["pod_present"][p])
        for z in range(nz):
            tot = 0; k = 0; parts = []
            for r in range(R):
                if not (pres >> r) & 1: continue
                k += 1
                cap = int(av[z, r]) if (zp[z] >> r) & 1 else 0
                v = int(req[r])
                if r == 0:
                    fits = v <= cap; vv = -(-v // 1000); cc = -(-cap // 1000)
                else:
                    fits = v <= cap; vv = v; cc = cap
                sc = 0 if (cc == 0 or not fits) else vv * 100 // cc
                parts.append(sc); tot += sc
            print("  zone", z, parts, "score", tot // k if k else None)
