#!/usr/bin/env python3
"""bench.py — headline benchmark of the batched Filter/Score engine.

Metric (BASELINE.json): pod x node Filter+Score evals/sec.  A "step" is one pass of the hot path (one spx_eval of the whole
plugin set) over one batch of synthetic pods against the node snapshot, with every input table already resident in HBM.
N=1 workload = BASELINE.json configs[1]: noderesources.Allocatable + trimaran.TargetLoadPacking, 10k nodes x 100k pods.

`--gpus N` with N > 1 runs in one of two shapes, same numbers either way:
  * launched bare (`python bench.py --gpus N`): ONE host process drives N devices through the C ABI's multi-device layer
    (spx_multi_*: an engine and a host thread per device, RCCL all-gather for the exchange) — the shape north_star names
    (a single Go scheduler, 8 GPUs);
  * launched by torch.distributed.run (WORLD_SIZE set): one rank per GPU, torch.distributed (RCCL) for barrier / exchange.
Pod rows are the sharded unit, node tables are replicated, the evaluation has no collective.  Weak-scaling workloads give
every GPU its own batch (config2: 100k pods per GPU); the workloads BASELINE quotes on 8 GPUs are strong-scaling: config4
shards 200k pods and config5 500k pods over the N GPUs.  The exchange (all-gather of the 20-byte per-pod decisions, and with
`--gather table` of one uint8 table) is measured outside the timed region and reported next to the kernel-only step.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)

WORKLOADS = {
    # n_pods: per GPU (weak) or for the whole job (strong); algorithmic bytes: node_row, pod_row, out per eval — SURVEY.md §8d
    "config2": dict(n_nodes=10_000, n_pods=100_000, plugins=("alloc", "tlp"), node_row=41, pod_row=8, out=2,
                    desc="noderesources.Allocatable + trimaran.TargetLoadPacking, 10k nodes x 100k pods"),
    "config2_lvrb": dict(n_nodes=10_000, n_pods=100_000, plugins=("alloc", "tlp", "lvrb"), node_row=90, pod_row=24, out=3,
                         desc="Allocatable + TargetLoadPacking + LoadVariationRiskBalancing, 10k x 100k"),
    # LowRiskOverCommitment alone: node_row = 4 int64 sums + the 7 LVRB columns (49 B), pod_row = 4 int64
    "config2_lroc": dict(n_nodes=10_000, n_pods=100_000, plugins=("lroc",), node_row=81, pod_row=32, out=1,
                         desc="trimaran.LowRiskOverCommitment, 10k nodes x 100k pods"),
    "config2_peaks": dict(n_nodes=10_000, n_pods=100_000, plugins=("peaks",), node_row=33, pod_row=8, out=1,
                          desc="trimaran.Peaks (Score + NormalizeScore), 10k nodes x 100k pods"),
    "config3": dict(n_nodes=5_000, n_pods=50_000, plugins=("nrt",), node_row=324, pod_row=100, out=2, strategy="LeastAllocated",
                    desc="noderesourcetopology Filter+Score (LeastAllocated), 5k nodes x 8 NUMA zones x 50k pods"),
    "config3_leastnuma": dict(n_nodes=5_000, n_pods=50_000, plugins=("nrt",), node_row=324, pod_row=100, out=2, strategy="LeastNUMANodes",
                              desc="noderesourcetopology Filter+Score (LeastNUMANodes), 5k nodes x 8 NUMA zones x 50k pods"),
    "config3_most": dict(n_nodes=5_000, n_pods=50_000, plugins=("nrt",), node_row=324, pod_row=100, out=2, strategy="MostAllocated",
                         desc="noderesourcetopology Filter+Score (MostAllocated), 5k nodes x 8 NUMA zones x 50k pods"),
    "config3_balanced": dict(n_nodes=5_000, n_pods=50_000, plugins=("nrt",), node_row=324, pod_row=100, out=2, strategy="BalancedAllocation",
                             desc="noderesourcetopology Filter+Score (BalancedAllocation), 5k nodes x 8 NUMA zones x 50k pods"),
    # the kernels' 8-slot instantiations: six NUMA-affine resources (cpu, memory, hugepages-2Mi, hugepages-1Gi, two extended resources);
    # node_row = 8 zones x 6 slots x 8 B + flags, pod_row = 8 containers x 6 x 8 B + header
    "config3_r8": dict(n_nodes=5_000, n_pods=50_000, plugins=("nrt",), node_row=452, pod_row=148, out=2, strategy="LeastAllocated", wide=True,
                       desc="noderesourcetopology Filter+Score (LeastAllocated), six resource slots, 5k nodes x 8 NUMA zones x 50k pods"),
    "config3_r8_balanced": dict(n_nodes=5_000, n_pods=50_000, plugins=("nrt",), node_row=452, pod_row=148, out=2, strategy="BalancedAllocation", wide=True,
                                desc="noderesourcetopology Filter+Score (BalancedAllocation), six resource slots, 5k x 8 x 50k"),
    "config4": dict(n_nodes=10_000, n_pods=200_000, plugins=("net",), node_row=4, pod_row=48, out=2, scaling="strong",
                    desc="networkaware NetworkOverhead (+TopologicalSort keys), 10k nodes x 3-tier topology x 200k pods sharded over the GPUs"),
    # out: SURVEY.md 8d counts 4 score tables + LVRB's + ONE status byte per eval = 6 (the engine keeps two status tables, NRT's
    # and NetworkOverhead's: the seventh byte it writes is not algorithmic)
    "config5": dict(n_nodes=20_000, n_pods=500_000, plugins=("cap", "alloc", "tlp", "lvrb", "nrt", "net"), node_row=405, pod_row=188, out=6,
                    strategy="LeastAllocated", scaling="strong",
                    desc="full profile: CapacityScheduling PreFilter + Allocatable + TLP + LVRB + NRT + NetworkOverhead, 20k nodes x 500k pods sharded over the GPUs"),
    # config5's one-GPU share of the 8-GPU job (62.5k of the 500k pods), as a single-device workload
    "config5_share": dict(n_nodes=20_000, n_pods=62_500, plugins=("cap", "alloc", "tlp", "lvrb", "nrt", "net"), node_row=405, pod_row=188, out=6,
                          strategy="LeastAllocated",
                          desc="full profile, 20k nodes x 62.5k pods (= one GPU's share of config5 on 8)"),
    # the reference's own benchmark shapes: BenchmarkTargetLoadPackingPlugin (targetloadpacking_test.go:283-384:
    # nodesNum / podsNum 100/1000, 1000/10000, 5000/30000; one pod per iteration, every node reporting CPU 0 Latest) and
    # BenchmarkNetworkOverhead* (networkoverhead_test.go:349-570: onlineboutique, 10 pods placed, 10 ... 10000 nodes)
    "ref_tlp_100": dict(n_nodes=100, n_pods=1_000, plugins=("tlp",), node_row=33, pod_row=8, out=1, ref_shape=True,
                        desc="BenchmarkTargetLoadPackingPlugin/100nodes: TLP, 100 nodes x 1000 pods"),
    "ref_tlp_1000": dict(n_nodes=1_000, n_pods=10_000, plugins=("tlp",), node_row=33, pod_row=8, out=1, ref_shape=True,
                         desc="BenchmarkTargetLoadPackingPlugin/1000nodes: TLP, 1000 nodes x 10000 pods"),
    "ref_tlp_5000": dict(n_nodes=5_000, n_pods=30_000, plugins=("tlp",), node_row=33, pod_row=8, out=1, ref_shape=True,
                         desc="BenchmarkTargetLoadPackingPlugin/5000nodes: TLP, 5000 nodes x 30000 pods"),
    "ref_net_1000": dict(n_nodes=1_000, n_pods=11_000, plugins=("net",), node_row=4, pod_row=48, out=2, pods_per_group=11,
                         desc="BenchmarkNetworkOverhead*/1000 nodes: AppGroups of 11 workloads (onlineboutique), 1000 nodes x 11000 pods"),
    "ref_net_10000": dict(n_nodes=10_000, n_pods=11_000, plugins=("net",), node_row=4, pod_row=48, out=2, pods_per_group=11,
                          desc="BenchmarkNetworkOverhead*/10000 nodes: AppGroups of 11 workloads (onlineboutique), 10000 nodes x 11000 pods"),
    "small": dict(n_nodes=1_000, n_pods=4_000, plugins=("alloc", "tlp"), node_row=41, pod_row=8, out=2,
                  desc="plumbing-sized Allocatable + TLP"),
    # plumbing-sized strong-scaling shapes: what tests/test_gpu_bench.py runs through `--devices 0,0` on a one-GPU box
    "small_net": dict(n_nodes=2_000, n_pods=8_000, plugins=("net",), node_row=4, pod_row=48, out=2, scaling="strong",
                      desc="plumbing-sized NetworkOverhead (+TopologicalSort keys), 2k nodes x 8k pods sharded over the GPUs"),
    "small_full": dict(n_nodes=2_000, n_pods=8_000, plugins=("cap", "alloc", "tlp", "lvrb", "nrt", "net"), node_row=405, pod_row=188, out=6,
                       strategy="LeastAllocated", scaling="strong",
                       desc="plumbing-sized full profile, 2k nodes x 8k pods sharded over the GPUs"),
    # a batch no rank count divides: 8191 = 2 x 4096 - 1 = 3 x 2731 - 2 (the last rank's shard is short)
    "small_full_ragged": dict(n_nodes=2_000, n_pods=8_191, plugins=("cap", "alloc", "tlp", "lvrb", "nrt", "net"), node_row=405, pod_row=188, out=6,
                              strategy="LeastAllocated", scaling="strong",
                              desc="plumbing-sized full profile, 2k nodes x 8191 pods sharded over the GPUs (ragged shards)"),
    # BASELINE.json configs[0]: noderesources.Allocatable (LeastAllocated) on 100 nodes x 1k pods — the reference's own CPU-runnable
    # case (test/integration/allocatable_test.go through hack/integration-test.sh); here the same shape through the C ABI
    "config1": dict(n_nodes=100, n_pods=1_000, plugins=("alloc",), node_row=16, pod_row=0, out=1,
                    desc="noderesources.Allocatable (LeastAllocated), 100 nodes x 1k pods (plumbing)"),
}
PID = {"alloc": 0, "tlp": 1, "lvrb": 2, "nrt": 3, "net": 4, "cap": 5, "lroc": 7, "peaks": 8}
SEQ_PATH = {1: "one-workgroup chain (Filter-less profile)", 2: "per-pod single-row launches replayed from a graph", 3: "cooperative persistent kernel"}


# kernel translation units per plugin key of WORKLOADS (csrc/)
KERNEL_FILES = {
    "alloc": ("kernels_trimaran.hip",), "tlp": ("kernels_trimaran.hip",), "lvrb": ("kernels_trimaran.hip",),
    "lroc": ("kernels_lroc.hip",), "peaks": ("kernels_peaks.hip", "kernels_profile.hip"),
    "nrt": ("kernels_nrt_fused.hip", "kernels_nrt_rank.hip", "kernels_nrt_fast.hip", "kernels_nrt.hip", "kernels_profile.hip"), "net": ("kernels_network.hip",),
    "cap": ("kernels_capacity.hip", "kernels_profile.hip"),
}


def kernel_source_hash(plugins=None) -> str:
    """identifies the kernels a profile was taken with (the GPU box has no .git): profiles/ entries carry it, and bench.py
    reports their counter figures only while it still matches.  It is a hash of the gfx950 machine code of the translation
    units the workload's sweep kernels are compiled from (build.device_code_hash over the in-tree objects), so it follows what
    the kernels ARE: an edit to a shared header or to another plugin's kernel leaves it alone, any change of the generated code
    moves it.  Falls back to hashing the sources ("src:" prefix, never equal to a stamped code hash) when the objects are absent."""
    from scheduler_plugins_amd import build as spx_build
    csrc = ROOT / "scheduler-plugins_amd" / "csrc"
    if plugins is None:
        names = sorted(f.name for f in csrc.glob("kernels_*.hip"))
    else:
        names = sorted({f for p in plugins for f in KERNEL_FILES[p]})
    try:
        return spx_build.device_code_hash(names)
    except (OSError, ValueError):
        h = hashlib.sha256()
        for name in names + ["spx_internal.h"]:
            h.update(name.encode())
            h.update((csrc / name).read_bytes())
        return "src:" + h.hexdigest()[:12]


def build_snapshot(hdr, w, n_pods, seed):
    from scheduler_plugins_amd import objects as O
    from scheduler_plugins_amd import synth
    n_nodes = w["n_nodes"]
    if "cap" in w["plugins"]:
        # quotas in proportion to the batch's requests (round 3): with the fixed-size quotas 89 % of a 62.5k-pod batch scheduled one
        # after the other ended over Max and the sequential commit mostly timed rejected pods
        snap = synth.full_snapshot(hdr, n_nodes, n_pods, seed=seed, quota_sized_for_batch=True)
        snap["nrt_params"] = O.nrt_params(hdr, O.Resources(), w["strategy"])
    elif "nrt" in w["plugins"]:
        wide = bool(w.get("wide"))
        snap = synth.nrt_snapshot(hdr, n_nodes, n_pods, seed=synth.SEED, wide=wide)
        if seed != synth.SEED:
            snap["pods"] = synth.synth_pods(hdr, n_pods, seed=seed, device_res=synth.RES_DEVICE, hugepage_res=synth.RES_HUGEPAGES_2MI,
                                            device2_res=synth.RES_DEVICE2 if wide else -1, hugepage2_res=synth.RES_HUGEPAGES_1GI if wide else -1)
        snap["nrt_params"] = O.nrt_params(hdr, O.Resources(), w["strategy"])
    elif "net" in w["plugins"]:
        snap = synth.network_snapshot(hdr, n_nodes, n_pods, seed=seed, **({"pods_per_group": w["pods_per_group"]} if "pods_per_group" in w else {}))
    else:
        snap = synth.trimaran_snapshot(hdr, n_nodes, n_pods, seed=synth.SEED, round_frac=w.get("round_frac", 0.0),
                                       with_node_pods="lroc" in w["plugins"])
        if seed != synth.SEED:
            snap["pods"] = synth.synth_pods(hdr, n_pods, seed=seed)
        if w.get("ref_shape"):  # every node reports one metric: CPU, Latest, 0 (targetloadpacking_test.go:325-334)
            snap["metrics"] = O.build_metrics_objects(hdr, n_nodes, {i: [("CPU", "Latest", 0.0)] for i in range(n_nodes)})
            snap["assigned"] = None
        if "peaks" in w["plugins"]:
            snap["power_models"] = synth.synth_power_models(hdr, n_nodes, synth.SEED)
    return snap


def load_tables(target, w, snap, rows=None):
    """target: Engine (rows = the slice of the batch this rank holds, None = all) or MultiEngine (shards itself)"""
    from scheduler_plugins_amd.engine import Engine
    pl = w["plugins"]
    sliced = isinstance(target, Engine) and rows is not None
    if any(p in pl for p in ("alloc", "tlp", "lvrb", "lroc", "peaks", "cap")):
        if sliced:
            target.upload_alloc_nodes(target.flatten_alloc_nodes(snap["nodes"], snap["rc"]))
            target.upload_trimaran_nodes(target.flatten_trimaran_nodes(snap["nodes"], snap["metrics"], snap.get("assigned")))
            target.upload_trimaran_pods(target.flatten_trimaran_pods(snap["pods"]), rows)
        else:
            target.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap.get("assigned"))
    if "peaks" in pl:
        target.load_peaks_objects(snap["nodes"], snap["metrics"], snap["power_models"], snap["pods"])
    if "lroc" in pl:
        if hasattr(target, "for_all"):
            target.for_all(lambda e: e.set_lroc())
        else:
            target.set_lroc()
        target.load_lroc_objects(snap["nodes"], snap["node_pods"], snap["pods"])
    if "nrt" in pl:
        if sliced:
            target.upload_nrt(target.flatten_nrt(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], snap["nrt_params"]), rows)
        else:
            target.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], snap["nrt_params"])
    if "net" in pl:
        if sliced:
            target.upload_network(target.flatten_network(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"]), rows)
        else:
            target.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
    if "cap" in pl:
        if sliced:
            target.upload_quota(target.flatten_quota(snap["pods"], snap["rc"], snap["quota"]), rows)
        else:
            target.load_quota_objects(snap["pods"], snap["rc"], snap["quota"])


def config5_leg(hdr, device, n_pods=8192):
    """A bounded leg of BASELINE config #5 inside the default (config #2) line, so that the full profile's numbers are driver-timed
    too: config #5's node count and plugin set, a batch of `n_pods` pods — sweep, decisions, and the one-pod-at-a-time cycle."""
    from scheduler_plugins_amd.engine import Engine
    w = dict(WORKLOADS["config5_share"], n_pods=n_pods)
    out = {"workload": f"full profile, {w['n_nodes']} nodes x {n_pods} pods (config #5's node count and plugins, a bounded batch)"}
    t0 = time.perf_counter()
    snap = build_snapshot(hdr, w, n_pods, synth_seed())
    out["synth_s"] = time.perf_counter() - t0
    mask = 0
    for p in w["plugins"]:
        mask |= 1 << PID[p]
    with Engine(device) as e:
        t0 = time.perf_counter()
        load_tables(e, w, snap)
        e.sync()
        out["flatten_upload_ms"] = (time.perf_counter() - t0) * 1e3
        try:  # the same snapshot through the library's one-call loaders (spx_load_*: what a cgo caller pays — no SoA columns on its side)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                e.load_c(snap, snap["nrt_params"])
                e.sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            out["load_c_ms"] = sorted(ts)[1]
            ts = []
            for _ in range(3):  # spx_load_profile: the same four loaders side by side on host threads of the library
                t0 = time.perf_counter()
                e.load_c(snap, snap["nrt_params"], concurrent=True)
                e.sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            out["load_profile_ms"] = sorted(ts)[1]
            # per loader (one more pass, each call bracketed by a sync) and, for the NRT loader — the largest — per stage
            per = {}
            for name, fn in (("trimaran", lambda: e._lib.spx_load_trimaran(e._h, snap["nodes"].ref(), snap["rc"].ref(), snap["pods"].ref(), snap["metrics"].ref(), snap["assigned"].ref())),
                             ("nrt", lambda: e._lib.spx_load_nrt(e._h, snap["nodes"].ref(), snap["nrt"].ref(), snap["rc"].ref(), snap["pods"].ref(), snap["nrt_params"].ref())),
                             ("network", lambda: e._lib.spx_load_network(e._h, snap["nodes"].ref(), snap["pods"].ref(), snap["appgroups"].ref(), snap["nettopo"].ref())),
                             ("quota", lambda: e._lib.spx_load_quota(e._h, snap["pods"].ref(), snap["rc"].ref(), snap["quota"].ref()))):
                e.sync()
                t0 = time.perf_counter()
                if fn() != 0:
                    raise RuntimeError(name)
                e.sync()
                per[name] = (time.perf_counter() - t0) * 1e3
            out["load_c_per_loader_ms"] = per
            out["load_c_nrt_stages_ms"] = e.last_load_nrt_ms()
        except Exception as ex:
            out["load_c_ms"] = {"error": repr(ex)[:200]}
        for _ in range(2):
            e.eval(mask)
        e.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            e.eval(mask)
        e.sync()
        out["sweep_ms"] = (time.perf_counter() - t0) * 1e3 / 5
        out["evals_per_sec"] = w["n_nodes"] * n_pods / (out["sweep_ms"] * 1e-3)
        uniq, copies = e.nrt_pod_classes()
        out["nrt_rows_evaluated"], out["nrt_rows_copied"] = uniq, copies
        e.set_option("NRT_POD_CLASSES", 0)  # the same sweep with every pod row evaluated (no representative rows + copies)
        e.eval(mask)
        e.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            e.eval(mask)
        e.sync()
        out["sweep_every_row_ms"] = (time.perf_counter() - t0) * 1e3 / 5
        e.set_option("NRT_POD_CLASSES", 1)
        e.eval(mask)
        e.eval_best(mask)
        e.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            e.eval(mask)
            e.eval_best(mask)
        e.sync()
        out["sweep_plus_argmax_ms"] = (time.perf_counter() - t0) * 1e3 / 5   # spx_eval + spx_eval_best: what spx_decide replaces
        e.decide(mask)
        e.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            e.decide(mask)
        e.sync()
        out["decide_ms"] = (time.perf_counter() - t0) * 1e3 / 5
        e.commit_sequential(mask, 0, min(n_pods, 256), want_ties=False)  # first call allocates
        t0 = time.perf_counter()
        node, _, _, _ = e.commit_sequential(mask, want_ties=False)
        dt = time.perf_counter() - t0
        out["sequential_commit_ms"] = dt * 1e3
        out["sequential_us_per_pod"] = dt * 1e6 / n_pods
        out["sequential_path"] = SEQ_PATH.get(e.commit_path(), "?")
        out["sequential_unschedulable"] = int((node < 0).sum())
        out["sequential_distinct_nodes"] = int(len(set(node.tolist())))
    return out


def workload_leg(hdr, device, name, steps):
    """One of BASELINE's other single-device workloads at its FULL size inside the default (config #2) line, so that the driver's
    run — not only the builder's `--workload` lines under profiles/ — times it: config #3 (NRT Filter+Score, 5k x 8 x 50k), config #4
    (NetworkOverhead, 10k x 200k), config #5's one-GPU share (full profile, 20k x 62.5k).  Tables resident, `steps` sweeps bracketed
    by HIP events on the engine's stream, a fraction of a second of GPU time each."""
    import torch
    from scheduler_plugins_amd.engine import Engine
    w = WORKLOADS[name]
    n_nodes, n_pods = w["n_nodes"], w["n_pods"]
    out = {"workload": w["desc"], "steps": steps}
    t0 = time.perf_counter()
    snap = build_snapshot(hdr, w, n_pods, synth_seed())
    out["synth_s"] = time.perf_counter() - t0
    mask = 0
    for p in w["plugins"]:
        mask |= 1 << PID[p]
    algo = n_nodes * w["node_row"] + n_pods * w["pod_row"] + n_nodes * n_pods * w["out"]
    stream = torch.cuda.current_stream(device)
    with Engine(device) as e:
        e.set_stream(stream.cuda_stream)
        t0 = time.perf_counter()
        load_tables(e, w, snap)
        e.sync()
        out["flatten_upload_ms"] = (time.perf_counter() - t0) * 1e3

        def timed():
            for _ in range(2):
                e.eval(mask)
            e.sync()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(stream)
            for _ in range(steps):
                e.eval(mask)
            ev1.record(stream)
            e.sync()
            return ev0.elapsed_time(ev1) / steps

        ms = timed()
        out.update({"kernel_ms": ms, "evals_per_sec": n_nodes * n_pods / (ms * 1e-3), "algorithmic_bytes": algo,
                    "achieved_GBs": algo / (ms * 1e-3) / 1e9, "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "kernel_source_hash": kernel_source_hash(w["plugins"])})
        if "nrt" in w["plugins"]:
            uniq, copies = e.nrt_pod_classes()
            out["nrt_rows_evaluated"], out["nrt_rows_copied"] = uniq, copies
            out["nrt_filter_path"] = e.nrt_filter_path()
            e.set_option("NRT_POD_CLASSES", 0)
            er = timed()
            e.set_option("NRT_POD_CLASSES", 1)
            out["every_row"] = {"kernel_ms": er, "frac": algo / (er * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "what": "SPX_OPT_NRT_POD_CLASSES off: every pod row evaluated, none copied"}
        counters = profile_counters(name, w["plugins"], out.get("nrt_rows_evaluated"), n_nodes)
        if counters:
            out["traffic"] = counters.get("traffic")
            out.update({k: v for k, v in counters.items() if k != "traffic"})
    return out


def synth_seed():
    from scheduler_plugins_amd import synth
    return synth.SEED


def delta_cycle(target, w, snap, hdr, mask, score_mask):
    """SURVEY 8d(ii) with deltas: one scheduling cycle against a snapshot that is already on the device — 1 % of the nodes changed
    (their trimaran and NRT rows replaced in place: spx_update_*_nodes), a NEW batch of pending pods of the same size (flattened and
    uploaded for every plugin of the workload), the sweep, the per-row argmax and the fetch of the decisions.  Wall clock."""
    from scheduler_plugins_amd import synth
    pl = w["plugins"]
    n_nodes, n_pods = w["n_nodes"], target.n_pods
    rng = np.random.default_rng(7)
    idx = np.sort(rng.choice(n_nodes, max(1, n_nodes // 100), replace=False))
    # the next cycle's pending batch (another seed: other requests, other AppGroup members)
    if "cap" in pl:
        pods = synth.full_snapshot(hdr, n_nodes, n_pods, seed=synth.SEED + 17, quota_sized_for_batch=True)["pods"]
    elif "nrt" in pl:
        pods = synth.synth_pods(hdr, n_pods, seed=synth.SEED + 17, device_res=synth.RES_DEVICE, hugepage_res=synth.RES_HUGEPAGES_2MI)
    elif "net" in pl:
        pods = synth.network_snapshot(hdr, n_nodes, n_pods, seed=synth.SEED + 17)["pods"]
    else:
        pods = synth.synth_pods(hdr, n_pods, seed=synth.SEED + 17)
    out = {}
    target.sync()
    t0 = time.perf_counter()
    if any(p in pl for p in ("tlp", "lvrb", "cap")):
        target.update_trimaran_node_rows(idx, target.flatten_trimaran_node_rows(snap["nodes"], snap["metrics"], snap.get("assigned"), idx))
    t0b = time.perf_counter()
    if "nrt" in pl:
        slots = target.nrt_soa["slots"]
        target.update_nrt_node_rows(idx, target.flatten_nrt_node_rows(snap["nodes"], snap["nrt"], slots, idx), int(slots.struct.n_res))
    t1 = time.perf_counter()
    if any(p in pl for p in ("alloc", "tlp", "lvrb", "cap")):
        target.load_trimaran_pods(pods)
    if "nrt" in pl:
        target.upload_nrt_pods(target.flatten_nrt_pods(pods, snap["rc"], slots), int(slots.struct.n_res))
    if "net" in pl:
        target.upload_network_pods(target.flatten_network_pods(pods, snap["appgroups"]))
    if "cap" in pl:
        target.upload_quota(target.flatten_quota(pods, snap["rc"], snap["quota"]))
    t2 = time.perf_counter()
    target.decide(score_mask)
    target.sync()
    t3 = time.perf_counter()
    target.best()
    t4 = time.perf_counter()
    out = {"ms": (t4 - t0) * 1e3, "node_delta_ms": (t1 - t0) * 1e3, "node_delta_trimaran_ms": (t0b - t0) * 1e3, "node_rows": int(len(idx)), "new_pods_ms": (t2 - t1) * 1e3,
           "decide_ms": (t3 - t2) * 1e3, "fetch_decisions_ms": (t4 - t3) * 1e3,
           "what": "1 % of the nodes' trimaran + NRT rows replaced in place (spx_update_*_nodes; both flattened for those nodes only), a new pending batch flattened and uploaded for every plugin, spx_decide (sweep + "
                   "per-row weighted argmax), D2H of the decisions"}
    return out


def cpu_baseline(spx, snap, e, plugins, budget_s: float):
    """Times the CPU oracle (C restatement of the reference's per-(pod,node) path — NOT the Go binary) on bounded samples of
    the same workload's pod rows, in three layouts: all host cores with pod rows split across threads (`value`: no per-pod
    join, the most favourable layout for the CPU), one thread, and the reference benchmark's own structure — one pod at a
    time, its node loop chunked over 16 workers that join per pod (targetloadpacking_test.go:369-405)."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import ctypes as C

    import pyoracle

    osnap = pyoracle.Snapshot(snap["nodes"], snap["pods"], rc=snap["rc"], metrics=snap.get("metrics"), assigned=snap.get("assigned"),
                              alloc_params=e.alloc_params, tlp_params=e.tlp_params, lvrb_params=e.lvrb_params,
                              nrt=snap.get("nrt"), nrt_params=snap.get("nrt_params"), appgroups=snap.get("appgroups"),
                              nettopo=snap.get("nettopo"), node_pods=snap.get("node_pods"), lroc_params=getattr(e, "lroc_params", None),
                              power_models=snap.get("power_models"))
    # what "all host cores" is on this box: the CPUs the process may run on, capped by the cgroup's CPU quota (the round-4 line said
    # "cores: 256" on a box whose cgroup grants 16 CPUs' worth of time — 256 runnable threads on 16 CPUs of quota)
    cores = pyoracle.usable_cpus()
    try:
        cpu_max = Path("/sys/fs/cgroup/cpu.max").read_text().strip()
    except OSError:
        cpu_max = None
    host_cpus = {"os_cpu_count": os.cpu_count(), "sched_getaffinity": len(os.sched_getaffinity(0)), "cgroup_cpu_max": cpu_max, "usable": cores}
    n_nodes = osnap.n_nodes

    def run_rows(rows: int, threads: int) -> float:
        t0 = time.perf_counter()
        for p in plugins:
            if p == 5:  # CapacityScheduling.PreFilter is per pod, not per (pod,node): negligible, not part of the CPU sample
                continue
            osnap.score_rows(p, 0, rows, threads=threads, want_raw=False, want_norm=True)
            if p in (3, 4):  # NodeResourceTopologyMatch / NetworkOverhead also have a Filter extension point
                osnap.filter_rows(p, 0, rows, threads=threads)
        return time.perf_counter() - t0

    def run_cycle(rows: int, workers: int) -> float:
        t0 = time.perf_counter()
        for p in plugins:
            if p in (4, 5):  # NetworkOverhead's PreFilter state is per pod: its reference benchmarks are single-goroutine loops
                continue
            pyoracle.lib().orc_cycle_rows(C.byref(osnap.struct), p, 0, rows, workers, None)
        return time.perf_counter() - t0

    def sized(fn, arg, probe_rows, share):
        probe_rows = max(1, min(osnap.n_pods, probe_rows))
        t = fn(probe_rows, arg)
        rows = int(max(probe_rows, min(osnap.n_pods, probe_rows / max(t, 1e-9) * budget_s * share)))
        t = fn(rows, arg)
        return rows, t

    rows, t = sized(run_rows, cores, 8 * cores, 0.5)
    out = {
        "value": rows * n_nodes / t, "unit": "evals/s", "cores": cores, "kind": "port", "host_cpus": host_cpus,
        "sample": f"{rows} pod rows x {n_nodes} nodes of the same snapshot, {len(plugins)} plugins, {t:.2f} s wall, pod rows split over "
                  f"{cores} threads (= usable CPUs: affinity capped by the cgroup quota); C restatement of the reference CPU path (oracle/), not the Go binary",
    }
    rows1, t1 = sized(run_rows, 1, 8, 0.25)
    out["single_thread"] = {"value": rows1 * n_nodes / t1, "cores": 1, "sample": f"{rows1} pod rows, {t1:.2f} s"}
    out["scaling_efficiency"] = out["value"] / (out["single_thread"]["value"] * cores)  # all-cores rate / (one-thread rate x usable CPUs)
    if any(p not in (4, 5) for p in plugins):
        rows16, t16 = sized(run_cycle, 16, 16, 0.25)
        out["reference_structure"] = {
            "value": rows16 * n_nodes / t16, "cores": 16, "sample": f"{rows16} pods one after the other, {t16:.2f} s",
            "what": "one pod at a time; its node loop in chunks of min(sqrt(N), N/16+1) claimed by 16 workers that join per pod, "
                    "NormalizeScore serial (workqueue.ParallelizeUntil as copied into targetloadpacking_test.go:386-405)"}
    return out


def profile_counters(workload: str, plugins, rows_evaluated=None, n_nodes=None):
    """counter figures of the committed rocprofv3 PMC passes (profiles/rNN/<workload>_traffic.json: separate --pmc runs, FETCH_SIZE
    corrected x2 for gfx950) — reported only when taken with the kernel sources this run was built from"""
    cands = sorted(ROOT.glob(f"profiles/r*/{workload}_traffic.json"))
    if not cands:
        return None
    d = json.loads(cands[-1].read_text())
    src = str(cands[-1].relative_to(ROOT))
    if d.get("kernel_source_hash") != kernel_source_hash(plugins):
        return {"traffic": None, "stale_profile": src}
    pmc = d.get("pmc_mean_per_dispatch", {})
    valu = None
    if pmc.get("SQ_BUSY_CYCLES") and pmc.get("SQ_ACTIVE_INST_VALU") is not None:
        # SQ_BUSY_CYCLES sums the chip's 32 SQ instances (4 per XCD: kernel cycles x 32, checked against GRBM_GUI_ACTIVE / 8 and the
        # kernel duration); the 1024 SIMDs issue one VALU instruction per 4 cycles each, and SQ_ACTIVE_INST_VALU counts in those
        # 4-cycle slots (it equals SQ_INSTS_VALU on these kernels): slots available = SQ_BUSY_CYCLES / 32 * 1024 / 4
        valu = pmc["SQ_ACTIVE_INST_VALU"] / (8.0 * pmc["SQ_BUSY_CYCLES"])
    out = {"traffic": d.get("traffic_bytes_per_launch"), "valu_busy_frac": valu, "source": src}
    if rows_evaluated and n_nodes and pmc.get("SQ_INSTS_VALU"):
        # instructions per (pod row evaluated, 64 nodes) — a wavefront's share of one pod — summed over the workload's sweep kernels, and the
        # time the vector ones alone take at one per 4 cycles per SIMD (1024 SIMDs, 2.4 GHz): progress on an issue-bound sweep counts in these
        cells = rows_evaluated * n_nodes / 64.0
        out["instr_per_cell"] = {"valu": pmc["SQ_INSTS_VALU"] / cells, "salu": pmc.get("SQ_INSTS_SALU", 0.0) / cells, "lds": pmc.get("SQ_INSTS_LDS", 0.0) / cells,
                                 "cell": "one evaluated pod row x 64 nodes (a wavefront's step)", "rows_evaluated": rows_evaluated,
                                 "valu_issue_bound_ms": pmc["SQ_INSTS_VALU"] * 4.0 / 1024.0 / 2.4e9 * 1e3}
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--plugins", default="", help="override the workload's plugin set, e.g. alloc or tlp,lvrb (experiments)")
    ap.add_argument("--round-frac", type=float, default=0.0, help="fraction of nodes with integer-valued metrics (tie stress)")
    ap.add_argument("--gather", default="best", choices=["none", "best", "table"],
                    help="N>1 only, measured OUTSIDE the timed region: all-gather of per-pod decisions and (table) of one score slab")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "copy"], help="single-process multi-device exchange (spx_multi)")
    ap.add_argument("--devices", default="", help="single-process multi-device mode: explicit device list, e.g. 0,0 with --transport copy "
                                                  "runs two ranks on one GPU (plumbing check of the sharded path on a one-GPU box)")
    ap.add_argument("--sweep-only", action="store_true", help="skip the full_cycle section (profiling runs: rocprofv3 counter passes crash in hipGraph capture)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="spx_set_option before the timed region, e.g. --opt NET_ALLOC_FUSED=0 (A/B experiments; repeatable)")
    ap.add_argument("--no-pod-classes", action="store_true", help="evaluate every pod row (SPX_OPT_NRT_POD_CLASSES / SPX_OPT_PEAKS_POD_CLASSES off): "
                    "by default a whole-batch NRT or Peaks sweep evaluates one row per class of pods with equal records and copies it")
    ap.add_argument("--no-every-row", action="store_true", help="skip the every_row section (the sweep once more with pod classes off): profiler passes "
                    "use it so that per-dispatch counter means describe the timed sweep only")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="ranks mode (torch.distributed.run): nccl = RCCL over xGMI (the driver's "
                    "scaling run); gloo = the same code path with the collectives on host tensors, for boxes with fewer GPUs than ranks (tests)")
    ap.add_argument("--rank-devices", default="", help="ranks mode: device of each local rank, e.g. 0,0 puts two ranks on device 0 (needs --dist-backend gloo: RCCL "
                    "refuses two ranks on one device)")
    ap.add_argument("--verify-gather", action="store_true", help="ranks mode, strong-scaling workloads: rank 0 also evaluates the whole batch in one engine and "
                    "compares the all-gathered decisions with it (gather.mismatches; small shapes — tests)")
    ap.add_argument("--no-config5-leg", action="store_true", help="default (config2) line only: skip the bounded full-profile leg (config5_leg)")
    ap.add_argument("--no-legs", action="store_true", help="default (config2) line only: skip the full-size config3 / config4 / config5_share legs")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU-oracle work for cpu_baseline (0 = skip)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rank_devices:  # (tests) several ranks per device
        local_rank = int(args.rank_devices.split(",")[local_rank])
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env > 1 and world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if args.devices:
        args.gpus = len(devices)
    elif args.gpus > torch.cuda.device_count() and not args.rank_devices:
        raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} device(s) visible")
    mode = "ranks" if world_env > 1 else ("multi" if args.gpus > 1 else "single")
    world = args.gpus
    torch.cuda.set_device(local_rank)
    dist = None
    if mode == "ranks":
        import torch.distributed as dist  # RCCL over xGMI
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        # every rank of the job must be in the communicator the exchange runs on — counted BY a collective, not read from the environment:
        # a rank that did not join (a device RCCL could not open) must stop the run here, not show up as a short all-gather later
        probe = torch.ones(1, dtype=torch.int64, device=f"cuda:{local_rank}" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(probe)
        coll_ranks = int(probe.item())
        if coll_ranks != args.gpus or dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the {args.dist_backend} communicator holds {coll_ranks} rank(s) (world size {dist.get_world_size()})")
    coll_dev = torch.device("cuda", local_rank) if args.dist_backend == "nccl" else torch.device("cpu")  # where the collectives' tensors live

    import scheduler_plugins_amd as spx
    from scheduler_plugins_amd import synth
    from scheduler_plugins_amd.engine import Engine, mask_of
    from scheduler_plugins_amd.multi import PEER_COPY, RCCL, MultiEngine

    w = dict(WORKLOADS[args.workload])
    if args.plugins:
        w["plugins"] = tuple(args.plugins.split(","))
        w["out"] = len(w["plugins"])
    if args.round_frac:
        w["round_frac"] = args.round_frac
    plugins = [PID[p] for p in w["plugins"]]
    mask = mask_of(*plugins)
    n_nodes = w["n_nodes"]
    strong = w.get("scaling") == "strong"
    n_pods_total = w["n_pods"] if strong else w["n_pods"] * world  # pods evaluated per step by the whole job
    hdr = spx.header()

    # ------------------------------------------------------------------ tables into HBM
    if mode == "multi":
        snap = build_snapshot(hdr, w, n_pods_total, synth.SEED)
        target = MultiEngine(devices, RCCL if args.transport == "rccl" else PEER_COPY)
        t_load = time.perf_counter()
        load_tables(target, w, snap)
        host_load = {"wall_ms": (time.perf_counter() - t_load) * 1e3, "per_rank_ms": [round(x, 3) for x in target.load_ms],
                     "what": "objects -> SoA -> HBM before the timed region: one host thread per rank flattens and uploads that rank's own pod rows "
                             "(a view of the pod object table) and the replicated node tables; CapacityScheduling's tables are flattened once"}
        e0 = target.engines[0]
        local_pods = max(e.n_pods for e in target.engines)
    else:
        target = e0 = Engine(local_rank)
        for kv in args.opt:  # (options read at upload time must be set before the first upload; the others are applied below)
            if kv.startswith(("ROW_ALIGN=", "NRT_RANK_NARROW=")):
                e0.set_option(kv.partition("=")[0], int(kv.partition("=")[2]))
        if strong and world > 1:  # every rank builds the same batch and keeps its shard
            snap = build_snapshot(hdr, w, n_pods_total, synth.SEED)
            from scheduler_plugins_amd import shard
            rows = shard.shard_rows(n_pods_total, world, rank)  # ceil(P/G) rows per rank: the one partition rule (= spx_multi_shard)
            load_tables(target, w, snap, rows)
        else:  # same node snapshot, an own pod batch per rank
            snap = build_snapshot(hdr, w, w["n_pods"] if not strong else n_pods_total, synth.SEED + 1000 * rank)
            load_tables(target, w, snap)
        local_pods = target.n_pods

    engines = target.engines if mode == "multi" else [e0]
    if args.no_pod_classes:
        for e in engines:
            e.set_option("NRT_POD_CLASSES", 0)
            e.set_option("PEAKS_POD_CLASSES", 0)
    for kv in args.opt:
        name, _, val = kv.partition("=")
        if name == "ROW_ALIGN" and mode != "multi":
            continue  # applied at engine creation
        for e in engines:
            e.set_option(name, int(val))
    pod_classes = {}
    for name, fn in (("nrt", e0.nrt_pod_classes), ("peaks", e0.peaks_pod_classes)):
        if name in w["plugins"]:
            uniq, copies = fn()
            pod_classes[name] = {"rows_evaluated": uniq, "rows_copied": copies, "enabled": not args.no_pod_classes}

    def barrier():
        if dist is not None:
            dist.barrier()
        if mode == "multi":
            target.sync()
        torch.cuda.synchronize()

    # bring the device(s) out of idle clocks with unrelated work (not steps of the workload): a fresh box runs its first
    # ~50 ms of kernels at low clocks, which would otherwise dominate short --steps runs
    for d in (sorted(set(devices)) if mode == "multi" else [local_rank]):
        spin = torch.empty(64 << 20, dtype=torch.float32, device=f"cuda:{d}")
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.15:
            for _ in range(8):
                spin.mul_(1.0001)
            torch.cuda.synchronize(d)
        del spin

    if mode == "multi":
        for _ in range(args.warmup):
            target.eval(mask)
        barrier()
        t0 = time.perf_counter()
        target.mark(0)
        for _ in range(args.steps):
            target.eval(mask)
        target.mark(1)
        barrier()
        elapsed = time.perf_counter() - t0
        kern_ms = target.marked_ms()[0] / args.steps  # slowest rank's HIP-event time over the timed region, per step
    else:
        # the engine launches on torch's current stream so that torch.cuda.Event brackets exactly its kernels
        tstream = torch.cuda.Stream(device=local_rank)
        torch.cuda.set_stream(tstream)
        target.set_stream(tstream.cuda_stream)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(args.warmup):
            target.eval(mask)
        target.sync()
        barrier()
        t0 = time.perf_counter()
        ev0.record(tstream)
        for _ in range(args.steps):
            target.eval(mask)
        ev1.record(tstream)
        target.sync()
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # average launch duration of the sweep measured with HIP events over the timed region itself (back-to-back
        # launches, sustained clocks); a plugin set evaluated by more than one kernel counts all of them as one launch
        kern_ms = ev0.elapsed_time(ev1) / args.steps

    # the same sweep with every pod row evaluated (no representative rows + copies), outside the timed region: how much of `value`
    # is the synthetic queue's repetitiveness — printed next to it in every line that uses pod classes
    every_row = None
    if mode == "single" and not args.no_pod_classes and not args.no_every_row and any(v["rows_copied"] > 0 for v in pod_classes.values()):
        try:
            for name in pod_classes:
                target.set_option("NRT_POD_CLASSES" if name == "nrt" else "PEAKS_POD_CLASSES", 0)
            n_er = max(3, min(args.steps, 10))
            target.eval(mask)
            target.sync()
            er0, er1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            er0.record(tstream)
            for _ in range(n_er):
                target.eval(mask)
            er1.record(tstream)
            target.sync()
            every_row = {"kernel_ms": er0.elapsed_time(er1) / n_er, "steps": n_er,
                         "what": "the same sweep with SPX_OPT_*_POD_CLASSES off: every pod row evaluated, none copied (HIP events, outside the timed region)"}
        except Exception as ex:
            every_row = {"error": repr(ex)[:200]}
        finally:
            for name in pod_classes:
                target.set_option("NRT_POD_CLASSES" if name == "nrt" else "PEAKS_POD_CLASSES", 1)
            target.eval(mask)
            target.sync()

    # ------------------------------------------------------------------ outside the timed region
    # SURVEY §8d(ii) "full-cycle ms": snapshot delta (host flatten + H2D of the SoA columns) + sweep + device-side
    # per-row argmax + D2H of the per-pod decisions — wall clock, once
    full_cycle = None
    score_mask = mask & ~(1 << 6)
    if mode == "single" and not args.plugins and not args.sweep_only and args.workload in ("config2", "config2_lvrb", "config5_share", "config4", "config3"):
        try:
            barrier()
            c0 = time.perf_counter()
            load_tables(target, w, snap)
            c1 = time.perf_counter()
            target.eval(mask)
            target.eval_best(score_mask)
            target.sync()
            c2 = time.perf_counter()
            target.best()
            c3 = time.perf_counter()
            full_cycle = {"ms": (c3 - c0) * 1e3, "flatten_upload_ms": (c1 - c0) * 1e3, "eval_argmax_ms": (c2 - c1) * 1e3,
                          "fetch_decisions_ms": (c3 - c2) * 1e3,
                          "what": "objects->SoA flatten + H2D, sweep, per-row weighted argmax, D2H of 20 B/pod decisions"}
            # decisions without tables (spx_decide: the argmax folded into the sweep where the profile allows)
            try:
                for _ in range(3):
                    target.decide(score_mask)
                target.sync()
                d0 = time.perf_counter()
                for _ in range(10):
                    target.decide(score_mask)
                target.sync()
                full_cycle["decide_ms"] = (time.perf_counter() - d0) * 1e3 / 10
                full_cycle["decide_what"] = ("sweep + per-row argmax, same decisions as eval_argmax.  Filter-less trimaran profiles: no score table written (argmax folded into "
                                             "the sweep); profiles with Filter plugins: their sweeps write their tables, Allocatable's feasibility-aware "
                                             "NormalizeScore is folded into the argmax kernel (k_decide_masked: no Allocatable table, status rows read once)")
            except Exception as ex:
                full_cycle["decide_error"] = repr(ex)[:200]
            if args.workload in ("config2", "config2_lvrb", "config5_share", "config3"):
                try:
                    first = delta_cycle(target, w, snap, hdr, mask, score_mask)  # first use allocates the staging buffers
                    full_cycle["delta_cycle"] = delta_cycle(target, w, snap, hdr, mask, score_mask)
                    full_cycle["delta_cycle"]["first_call_ms"] = first["ms"]
                    load_tables(target, w, snap)  # back to the workload's own batch for what follows
                except Exception as ex:
                    full_cycle["delta_cycle"] = {"error": repr(ex)[:200]}
            if args.workload in ("config2", "config2_lvrb", "config5_share"):
                # the same pods scheduled strictly one after the other, each seeing the commits before it (upstream's
                # semantics; inherently sequential, one workgroup): spx_commit_sequential
                c4 = time.perf_counter()
                seq_node, _, _, _ = target.commit_sequential(mask, want_ties=False)
                c5 = time.perf_counter()
                full_cycle["sequential_commit_ms"] = (c5 - c4) * 1e3
                full_cycle["sequential_pods_per_s"] = local_pods / (c5 - c4)
                full_cycle["sequential_distinct_nodes"] = int(len(set(seq_node.tolist())))
                full_cycle["sequential_unschedulable"] = int((seq_node < 0).sum())
                full_cycle["sequential_us_per_pod"] = (c5 - c4) * 1e6 / local_pods
                full_cycle["sequential_path"] = SEQ_PATH.get(target.commit_path(), "?")
                full_cycle["sequential_what"] = ("one workgroup carrying trimaran's bind-time state in registers (k_commit_trimaran_reg)" if args.workload != "config5_share" else
                                                 "one cooperative persistent launch: a workgroup per 256 nodes keeps NRT zones / TLP columns in registers, two granule "
                                                 "exchanges per pod (feasible-set extremes, weighted argmax), Reserve bookkeeping (NRT assumed resources, AppGroup "
                                                 "scheduled list, ElasticQuota used, trimaran cache) applied in place — k_commit_coop; falls back to per-pod launches "
                                                 "when the profile does not fit (sequential_path)")
        except Exception as ex:
            full_cycle = {"error": repr(ex)[:200]}

    # config #4 names "NodeNetworkCost + TopologicalSort": the queue sort of the whole batch (one device; 16-byte keys), device time
    sort_info = None
    if "net" in w["plugins"] and rank == 0:
        try:
            q_pods = snap["pods"]
            if mode == "multi":
                target.sort_queue(q_pods)
                ms = target.engines[0].last_eval_ms()
            elif strong and world > 1:
                f = target.flatten_network(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
                target.sort_queue(q_pods, topo_order=f["cols"]["topo_order"])
                ms = target.last_eval_ms()
            else:
                target.sort_queue(q_pods)
                ms = target.last_eval_ms()
            sort_info = {"queue_sort_ms": ms, "n_keys": int(q_pods.struct.n_pods),
                         "what": "spx_sort_keys: TopologicalSort order of the whole pending queue (two stable radix sorts of pod indices), HIP-event time"}
        except Exception as ex:
            sort_info = {"error": repr(ex)[:200]}

    # the exchange step of the sharded path, reported separately: per-pod decisions, optionally one table
    gather_info = None
    if world > 1 and args.gather != "none":
        try:
            if mode == "multi":
                target.eval(mask)
                target.eval_best(score_mask)
                barrier()
                t1 = time.perf_counter()
                target.gather_best()
                best_wall = (time.perf_counter() - t1) * 1e3
                gather_info = {"best_ms": best_wall, "best_device_ms": target.last_ms()[1], "bytes_per_rank": int(local_pods) * 20,
                               "transport": args.transport, "host": "one process, spx_multi_gather_best (all-gather + one D2H)"}
                barrier()
                t1 = time.perf_counter()
                target.eval(mask)
                target.eval_best(score_mask)
                target.gather_best()
                gather_info["step_plus_gather_ms"] = (time.perf_counter() - t1) * 1e3
                gather_info["step_plus_gather_what"] = "sweep + per-row argmax + all-gather of decisions + D2H, wall clock (kernel-only step: ms_per_step)"
                if args.gather == "table":
                    p0 = plugins[-1] if plugins[-1] <= 4 else plugins[0]
                    target.bind_global_table(p0)
                    target.eval(1 << p0)
                    barrier()
                    t1 = time.perf_counter()
                    target.allgather_table(p0)
                    barrier()
                    gather_info.update({"table_ms": (time.perf_counter() - t1) * 1e3, "table_device_ms": target.last_ms()[1],
                                        "table_bytes": int(-(-n_pods_total // world) * world * e0.score_table(p0)[1])})
            else:
                from scheduler_plugins_amd import shard
                target.eval_best(score_mask)
                node, score, ties, feas = target.best()
                barrier()
                t1 = time.perf_counter()
                if strong:  # shards of one batch (ceil(P/G) rows per rank, the last one short: shard.shard_rows)
                    gathered = shard.gather_best(dist, coll_dev, node, score, ties, feas, n_pods_total)
                else:
                    gathered = shard.gather_best(dist, coll_dev, node, score, ties, feas, local_pods * world)
                barrier()
                gather_info = {"best_ms": (time.perf_counter() - t1) * 1e3, "bytes_per_rank": int(local_pods) * 20,
                               "host": "one process per GPU, torch.distributed all_gather_into_tensor", "backend": args.dist_backend,
                               "note": "no N > 1 hardware figure has been measured by the builder: one-GPU boxes only (two ranks on one device run over gloo)"}
                if args.verify_gather and strong and rank == 0:
                    # the assembled decisions against ONE engine holding the whole batch (small shapes: tests of the partition rule)
                    with Engine(local_rank) as whole:
                        load_tables(whole, w, snap)
                        whole.eval(mask)
                        whole.eval_best(score_mask)
                        whole.sync()
                        want = whole.best()
                    gather_info["verified_rows"] = int(len(want[0]))
                    gather_info["mismatches"] = int(sum(int((np.asarray(g) != np.asarray(x)).sum()) for g, x in zip(gathered, want)))
                if args.gather == "table":
                    p0 = plugins[-1] if plugins[-1] <= 4 else plugins[0]
                    ptr, stride, rows_t = target.score_table(p0)
                    slab = torch.empty((rows_t, stride), dtype=torch.uint8, device=f"cuda:{local_rank}")
                    target.bind_score_table(p0, slab.data_ptr(), stride, rows_t)
                    target.eval(1 << p0)
                    target.sync()
                    barrier()
                    t1 = time.perf_counter()
                    full = shard.gather_table(dist, slab if args.dist_backend == "nccl" else slab.cpu())
                    barrier()
                    gather_info.update({"table_ms": (time.perf_counter() - t1) * 1e3, "table_bytes": int(full.numel())})
                    del full
        except Exception as ex:  # never lose the bench line to the optional exchange measurement
            gather_info = {"error": repr(ex)[:300]}

    evals_per_step = n_nodes * n_pods_total
    value = evals_per_step * args.steps / elapsed
    # roofline of the slowest rank's launch: the algorithmic bytes of ITS rows over its kernel time
    algo_bytes = n_nodes * w["node_row"] + local_pods * w["pod_row"] + n_nodes * local_pods * w["out"]
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9

    counters = profile_counters(args.workload, w["plugins"], pod_classes.get("nrt", {}).get("rows_evaluated") if pod_classes.get("nrt", {}).get("enabled") else
                                (local_pods if "nrt" in w["plugins"] else None), n_nodes) if not args.plugins else None
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": counters.get("traffic") if counters else None,
                "kernel": {"nrt": ("spx::k_nrt_fused (Filter + LeastAllocated Score in one launch) + k_nrt_fused_pack + k_rows_expand" if w.get("strategy") == "LeastAllocated" else
                                   "spx::k_nrt_filter_rank (Filter launch) + spx::k_nrt_fast (Score launch), both counted, + k_rows_expand"), "net": "spx::k_net_cls",
                           "lroc": "spx::k_lroc_fast (float32 quotient on exact float64 numerator/denominator, float64 fallback; VALU-bound)",
                           "peaks": "spx::k_peaks_minmax_est + k_peaks_fix_minmax + k_peaks_write_est + k_peaks_fix_write (float32 interval per cell, the float64 "
                                    "division + exp only for the listed cells; issue-bound) over one row per distinct pod cpu request + spx::k_rows_expand "
                                    "(config.pod_classes); SPX_OPT_PEAKS_ESTIMATE=0: spx::k_peaks<min/max pass> + spx::k_peaks<write pass>",
                           "cap": "full profile: k_quota, k_nrt_fused, k_net_cls (+ masked Allocatable), k_tlp_fast2, k_lvrb_fast, k_rows_expand"}.get(
                    w["plugins"][0], "spx::k_tlp_fast2 (Allocatable+TLP)" + (" + spx::k_lvrb_fast" if "lvrb" in w["plugins"] else "")),
                "kernel_ms": kern_ms, "algorithmic_bytes": algo_bytes, "frac_of_measured_copy_ceiling_6.29TBs": achieved / 6290.0,
                "kernel_source_hash": kernel_source_hash(w["plugins"])}
    if counters:
        roofline.update({k: v for k, v in counters.items() if k != "traffic"})
        roofline["valu_busy_what"] = "VALU issue slots used, summed over the workload's sweep kernels: SQ_ACTIVE_INST_VALU / (8 * SQ_BUSY_CYCLES) (32 SQ instances, 1024 SIMDs, one VALU instruction per SIMD per 4 cycles); committed PMC pass — why an HBM fraction is low when it is (issue-bound sweep)"
    out = {
        "metric": "pod_x_node_filter_score_evals_per_sec",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        # the arithmetic the sweep computes in (results are bit-exact against the reference's float64 / int64 either way):
        # TLP/LVRB float32 with a per-cell exactness proof and a float64 fallback, NRT float64 (exact integers), NetworkOverhead int32
        "dtype": {"nrt": "f64", "net": "i32", "cap": "f32+f64+i32", "lroc": "f64", "peaks": "f64"}.get(w["plugins"][0], "f32+f64"),
        "data": "synthetic",
        "config": {"workload": w["desc"], "n_nodes": n_nodes, "n_pods_per_step": n_pods_total, "n_pods_slowest_rank": int(local_pods),
                   "plugins": list(w["plugins"]),
                   "host": {"single": "one process, one device (`--gpus 1` IS this path: the N=1 point of a scaling curve is this line's value)", "multi": f"one process driving {world} devices through spx_multi (C ABI)",
                            "ranks": f"{world} processes, one per device (torch.distributed.run)"}[mode],
                   "sharding": "pod rows per device, node tables replicated, no data-path collective",
                   "result_tables": "uint8 [pods][nodes] per plugin, resident in HBM",
                   **({"options": args.opt} if args.opt else {}),
                   **({"pod_classes": dict(pod_classes, what="rows of pods whose records agree in everything the plugin reads are evaluated once and "
                                                              "copied (device 0's share; --no-pod-classes evaluates every row)")} if pod_classes else {})},
        "roofline": roofline,
        "kernel_evals_per_sec": n_nodes * local_pods / (kern_ms * 1e-3),
    }
    if every_row is not None:
        out["every_row"] = every_row
    if full_cycle is not None:
        out["full_cycle"] = full_cycle
    if mode == "multi":
        out["host_load"] = host_load
        # how many ranks the exchange's communicator has, as RCCL itself reports it (0: peer copies); fewer than the devices of an RCCL run is an error
        out["collective"] = {"transport": args.transport, "rccl_ranks": target.rccl_ranks()}
        if args.transport == "rccl" and out["collective"]["rccl_ranks"] != world:
            raise SystemExit(f"--gpus {world} over RCCL but the communicator holds {out['collective']['rccl_ranks']} rank(s)")
    elif mode == "ranks":
        out["collective"] = {"backend": args.dist_backend, "ranks": int(dist.get_world_size()), "rccl_ranks": coll_ranks if args.dist_backend == "nccl" else 0,
                             "ranks_counted_by_all_reduce": coll_ranks,
                             "what": "torch.distributed process group the decisions / table are all-gathered on (nccl = RCCL over xGMI)"}
    if world > 1:
        out["n_gt_1_hardware"] = ("no figure on more than one MI355X exists for this repository: the build box has one device; lines with several ranks "
                                  "on one device exercise the code path, not the scaling") if (len(set(devices)) == 1 or torch.cuda.device_count() < world) else "one rank per device"
    if gather_info is not None:
        out["gather"] = gather_info
    if sort_info is not None:
        out["topological_sort"] = sort_info
    if rank == 0 and world == 1 and mode == "single" and args.workload == "config2" and not args.plugins and not args.sweep_only and not args.no_config5_leg:
        try:
            out["config5_leg"] = config5_leg(hdr, local_rank)
        except Exception as ex:
            out["config5_leg"] = {"error": repr(ex)[:300]}
    if rank == 0 and world == 1 and mode == "single" and args.workload == "config2" and not args.plugins and not args.sweep_only and not args.no_legs:
        # BASELINE's other single-device workloads at full size, a few steps each (the driver's default run times them too)
        for leg, name, n in (("config3_leg", "config3", 10), ("config4_leg", "config4", 5), ("config5_share_leg", "config5_share", 5)):
            try:
                out[leg] = workload_leg(hdr, local_rank, name, n)
            except Exception as ex:
                out[leg] = {"error": repr(ex)[:300]}
    if rank == 0 and args.cpu_budget > 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(spx, snap, e0, plugins, args.cpu_budget)
    target.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
