import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import numpy as np
import pyoracle as oracle
from scheduler_plugins_amd import synth
from scheduler_plugins_amd.engine import Engine, mask_of
hdr = oracle.header()
n_nodes, n_pods = 257, 64
snap = synth.trimaran_snapshot(hdr, n_nodes, n_pods, seed=2, round_frac=0.1)
snap["power_models"] = synth.synth_power_models(hdr, n_nodes, 2)
s = oracle.Snapshot(snap["nodes"], snap["pods"], metrics=snap["metrics"], power_models=snap["power_models"])
raw_w, norm_w = s.score_rows(8)
with Engine(0) as e:
    e.load_peaks_objects(snap["nodes"], snap["metrics"], snap["power_models"], snap["pods"])
    e.eval(mask_of(8)); e.sync()
    got = e.all_scores(8).astype(np.int64)
    raw_g = np.stack([e.raw(8, r) for r in range(n_pods)])
print("raw maxabs diff", np.abs(raw_g - raw_w).max(), "raw max", np.abs(raw_w).max())
bad = np.argwhere(np.abs(got - norm_w) > 1)
print("bad cells", len(bad), "of", got.size)
for r, c in bad[:8]:
    print(r, c, "got", got[r, c], "want", norm_w[r, c], "raw", raw_w[r, c], raw_g[r, c], "row min/max", raw_w[r].min(), raw_w[r].max(), raw_g[r].min(), raw_g[r].max())
print("rows with bad:", sorted(set(bad[:, 0].tolist()))[:20])
print("neg raws:", (raw_w < 0).sum())
