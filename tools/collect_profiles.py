#!/usr/bin/env python3
"""Copies the judged summaries out of gpurun_out/prof_<workload>/ into profiles/<round>/ :
kernel stats CSV, a PMC summary per kernel, the bench line printed under rocprofv3, and a traffic JSON
(WRITE_SIZE + 2*FETCH_SIZE per launch of the dominant kernel, per MI355X_MICROARCH.md's gfx950 correction)."""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
only = set(sys.argv[2:])  # optional: the workloads to collect (default: every gpurun_out/prof_* directory)
out = ROOT / "profiles" / rnd
out.mkdir(parents=True, exist_ok=True)
for d in sorted((ROOT / "gpurun_out").glob("prof_*")):
    w = d.name[len("prof_"):]
    if only and w not in only:
        continue
    stats = d / "trace" / "t_kernel_stats.csv"
    if not stats.exists():
        continue
    shutil.copy(stats, out / f"{w}_kernel_stats.csv")
    rows = list(csv.DictReader(open(stats)))
    # the dominant SWEEP kernel: one-off kernels of the same command (the sequential commit loop, the argmax, prepare
    # steps, copies) are listed in the stats CSV but are not what bench.py's roofline line describes
    sweep = [r for r in rows if any(k in r["Name"] for k in ("k_tlp_fast", "k_lvrb_fast", "k_trimaran<", "k_nrt", "k_net", "k_alloc_masked", "k_lroc", "k_peaks<", "k_peaks_minmax", "k_peaks_write", "k_peaks_fix", "k_quota", "k_rows_expand"))]
    dom = max(sweep or rows, key=lambda r: float(r["TotalDurationNs"]))
    # kernels dispatched once per upload rather than once per step (the packed Score's table of exceptions, derived node columns)
    # are not part of a launch: only those dispatched at least half as often as the dominant one count
    sweep = [r for r in sweep if int(r["Calls"]) * 2 >= int(dom["Calls"])]
    bench_line = [l for l in open(d / "trace.log") if l.startswith("{")]
    summ = {"workload": w, "dominant_kernel": dom["Name"], "rocprof_avg_ns": float(dom["AverageNs"]), "calls": int(dom["Calls"])}
    if bench_line:
        b = json.loads(bench_line[-1])
        summ["bench_kernel_ms_under_rocprof"] = b["roofline"]["kernel_ms"]
        summ["kernel_source_hash"] = b["roofline"].get("kernel_source_hash")  # bench.py reports these counters only while it matches
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for pm in sorted(d.glob("pmc*/p_counter_collection.csv")):
        for r in csv.DictReader(open(pm)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # one bench "launch" = one dispatch of every sweep kernel of the workload (bench.py counts a plugin set evaluated by more than
    # one kernel as one launch), so the per-launch counters are the sums of the kernels' per-dispatch means; the PMC passes run
    # with --sweep-only, i.e. these kernels are dispatched once per step and nowhere else
    names = [r["Name"] for r in (sweep or [dom]) if r["Name"] in acc]
    counters = collections.defaultdict(float)
    per_kernel = {}
    for k in names:
        means = {c: sum(v) / len(v) for c, v in acc[k].items()}
        per_kernel[k] = means
        for c, v in means.items():
            counters[c] += v
    counters = dict(counters)
    summ["sweep_kernels"] = names
    summ["pmc_mean_per_dispatch"] = counters
    summ["pmc_per_kernel"] = per_kernel
    if "WRITE_SIZE" in counters and "FETCH_SIZE" in counters:
        summ["write_bytes"] = counters["WRITE_SIZE"] * 1024
        summ["fetch_bytes_corrected"] = counters["FETCH_SIZE"] * 1024 * 2
        summ["traffic_bytes_per_launch"] = summ["write_bytes"] + summ["fetch_bytes_corrected"]
        summ["note"] = ("WRITE_SIZE/FETCH_SIZE are KiB per dispatch from separate --pmc passes, summed over the workload's sweep kernels; "
                        "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)")
    (out / f"{w}_traffic.json").write_text(json.dumps(summ, indent=1) + "\n")
    print(w, dom["Name"][:60], f"avg {float(dom['AverageNs'])/1e6:.3f} ms", "traffic", summ.get("traffic_bytes_per_launch"))
