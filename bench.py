#!/usr/bin/env python3
"""bench.py — headline benchmark of the batched Filter/Score engine.

Metric (BASELINE.json): pod x node Filter+Score evals/sec.  A "step" is one pass of the hot path
(one spx_eval of the whole plugin set) over one batch of synthetic pods against the node snapshot,
with every input table already resident in HBM.  N=1 workload = BASELINE.json configs[1]:
noderesources.Allocatable + trimaran.TargetLoadPacking, 10k nodes x 100k pods.

N>1 (launched by torch.distributed.run, one rank per GPU): pod rows are the sharded unit — every rank
evaluates its own 100k-pod batch against the replicated node tables (weak scaling), no data-path
collective; `value` = all ranks' evals / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
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
    # name: (n_nodes, n_pods per GPU, plugins, algorithmic bytes: node_row, pod_row, out per eval) — SURVEY.md §8d
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
    "config4": dict(n_nodes=10_000, n_pods=200_000, plugins=("net",), node_row=4, pod_row=48, out=2,
                    desc="networkaware NetworkOverhead (+TopologicalSort keys), 10k nodes x 3-tier topology x 200k pods"),
    "config5": dict(n_nodes=20_000, n_pods=62_500, plugins=("cap", "alloc", "tlp", "lvrb", "nrt", "net"), node_row=405, pod_row=188, out=7,
                    strategy="LeastAllocated",
                    desc="full profile: CapacityScheduling PreFilter + Allocatable + TLP + LVRB + NRT + NetworkOverhead, 20k nodes x 62.5k pods per GPU (500k pods on 8)"),
    "small": dict(n_nodes=1_000, n_pods=4_000, plugins=("alloc", "tlp"), node_row=41, pod_row=8, out=2,
                  desc="plumbing-sized Allocatable + TLP"),
}


def cpu_baseline(spx, snap, e, plugins, budget_s: float):
    """Times the CPU oracle (C restatement of the reference's per-(pod,node) path — NOT the Go binary)
    on a bounded sample of the same workload's pod rows, all host cores, rows split across threads."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import pyoracle

    osnap = pyoracle.Snapshot(snap["nodes"], snap["pods"], rc=snap["rc"], metrics=snap.get("metrics"), assigned=snap.get("assigned"),
                              alloc_params=e.alloc_params, tlp_params=e.tlp_params, lvrb_params=e.lvrb_params,
                              nrt=snap.get("nrt"), nrt_params=snap.get("nrt_params"), appgroups=snap.get("appgroups"),
                              nettopo=snap.get("nettopo"), node_pods=snap.get("node_pods"), lroc_params=getattr(e, "lroc_params", None),
                              power_models=snap.get("power_models"))
    cores = os.cpu_count() or 1
    n_nodes = osnap.n_nodes

    def run(rows: int) -> float:
        t0 = time.perf_counter()
        for p in plugins:
            if p == 5:  # CapacityScheduling.PreFilter is per pod, not per (pod,node): negligible, not part of the CPU sample
                continue
            osnap.score_rows(p, 0, rows, threads=cores, want_raw=False, want_norm=True)
            if p in (3, 4):  # NodeResourceTopologyMatch / NetworkOverhead also have a Filter extension point
                osnap.filter_rows(p, 0, rows, threads=cores)
        return time.perf_counter() - t0
        return time.perf_counter() - t0

    probe_rows = min(osnap.n_pods, 8 * cores)
    t = run(probe_rows)
    rate = probe_rows / max(t, 1e-9)
    rows = int(max(probe_rows, min(osnap.n_pods, rate * budget_s)))
    t = run(rows)
    return {
        "value": rows * n_nodes / t, "unit": "evals/s", "cores": cores, "kind": "port",
        "sample": f"{rows} pod rows x {n_nodes} nodes of the same snapshot, {len(plugins)} plugins, {t:.2f} s wall; "
                  "C restatement of the reference CPU path (oracle/), not the Go binary",
    }


def measured_traffic(workload: str):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN/<workload>_traffic.json;
    collected by tools/prof1.sh in separate --pmc runs, FETCH_SIZE corrected x2 for gfx950)."""
    cands = sorted(ROOT.glob(f"profiles/r*/{workload}_traffic.json"))
    if not cands:
        return None, None
    d = json.loads(cands[-1].read_text())
    return d.get("traffic_bytes_per_launch"), str(cands[-1].relative_to(ROOT))


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
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU-oracle work for cpu_baseline (0 = skip)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # RCCL over xGMI
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import scheduler_plugins_amd as spx
    from scheduler_plugins_amd import synth
    from scheduler_plugins_amd import objects as O
    from scheduler_plugins_amd.engine import ALLOCATABLE, LVRB, NRT, TLP, Engine, mask_of

    w = dict(WORKLOADS[args.workload])
    if args.plugins:
        w["plugins"] = tuple(args.plugins.split(","))
        w["out"] = len(w["plugins"])
    pid = {"alloc": ALLOCATABLE, "tlp": TLP, "lvrb": LVRB, "nrt": NRT, "net": 4, "cap": 5, "lroc": 7, "peaks": 8}
    plugins = [pid[p] for p in w["plugins"]]
    mask = mask_of(*plugins)
    n_nodes, n_pods = w["n_nodes"], w["n_pods"]

    hdr = spx.header()
    # every rank: same node snapshot, its own pod batch (seeded by rank)
    e = Engine(local_rank)
    if "cap" in w["plugins"]:
        snap = synth.full_snapshot(hdr, n_nodes, n_pods, seed=synth.SEED + 1000 * rank)
        snap["nrt_params"] = O.nrt_params(hdr, O.Resources(), w["strategy"])
        e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], snap["nrt_params"])
        e.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
        e.load_quota_objects(snap["pods"], snap["rc"], snap["quota"])
    elif "nrt" in w["plugins"]:
        snap = synth.nrt_snapshot(hdr, n_nodes, n_pods, seed=synth.SEED)
        if rank:
            snap["pods"] = synth.synth_pods(hdr, n_pods, seed=synth.SEED + 1000 * rank, device_res=synth.RES_DEVICE,
                                            hugepage_res=synth.RES_HUGEPAGES_2MI)
        snap["nrt_params"] = O.nrt_params(hdr, O.Resources(), w["strategy"])
        e.load_nrt_objects(snap["nodes"], snap["nrt"], snap["rc"], snap["pods"], snap["nrt_params"])
    elif "net" in w["plugins"]:
        snap = synth.network_snapshot(hdr, n_nodes, n_pods, seed=synth.SEED + 1000 * rank)
        e.load_network_objects(snap["nodes"], snap["pods"], snap["appgroups"], snap["nettopo"])
    else:
        snap = synth.trimaran_snapshot(hdr, n_nodes, n_pods, seed=synth.SEED, round_frac=args.round_frac, with_node_pods="lroc" in w["plugins"])
        if rank:
            snap["pods"] = synth.synth_pods(hdr, n_pods, seed=synth.SEED + 1000 * rank)
        e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
        if "peaks" in w["plugins"]:
            snap["power_models"] = synth.synth_power_models(hdr, n_nodes, synth.SEED)
            e.load_peaks_objects(snap["nodes"], snap["metrics"], snap["power_models"], snap["pods"])
        if "lroc" in w["plugins"]:
            e.set_lroc()
            e.load_lroc_objects(snap["nodes"], snap["node_pods"], snap["pods"])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # the engine launches on torch's current stream so that torch.cuda.Event brackets exactly its kernels
    tstream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(tstream)
    e.set_stream(tstream.cuda_stream)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # bring the device out of its idle clocks with unrelated work (not steps of the workload): a fresh box runs its first
    # ~50 ms of kernels at low clocks, which would otherwise dominate short --steps runs
    spin = torch.empty(64 << 20, dtype=torch.float32, device=f"cuda:{local_rank}")
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.15:
        for _ in range(8):
            spin.mul_(1.0001)
        torch.cuda.synchronize()
    del spin
    for _ in range(args.warmup):
        e.eval(mask)
    e.sync()
    barrier()
    t0 = time.perf_counter()
    ev0.record(tstream)
    for _ in range(args.steps):
        e.eval(mask)
    ev1.record(tstream)
    e.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # average launch duration of the sweep measured with HIP events over the timed region itself (back-to-back
    # launches, sustained clocks); a plugin set evaluated by more than one kernel counts all of them as one launch
    kern_ms = ev0.elapsed_time(ev1) / args.steps

    # SURVEY §8d(ii) "full-cycle ms": snapshot delta (host flatten + H2D of the SoA columns) + sweep + device-side
    # per-row argmax + D2H of the per-pod decisions — measured once, outside the timed region, wall clock
    full_cycle = None
    if args.workload in ("config2", "config2_lvrb") and not args.plugins:
        try:
            barrier()
            c0 = time.perf_counter()
            e.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap["assigned"])
            c1 = time.perf_counter()
            e.eval(mask)
            e.eval_best(mask)
            e.sync()
            c2 = time.perf_counter()
            e.best()
            c3 = time.perf_counter()
            full_cycle = {"ms": (c3 - c0) * 1e3, "flatten_upload_ms": (c1 - c0) * 1e3, "eval_argmax_ms": (c2 - c1) * 1e3,
                          "fetch_decisions_ms": (c3 - c2) * 1e3,
                          "what": "objects->SoA flatten + H2D, sweep, per-row weighted argmax, D2H of 32 B/pod decisions"}
            # decisions without tables (spx_decide: the argmax folded into the sweep), HIP-event time of the launch
            try:
                for _ in range(3):
                    e.decide(mask)
                e.sync()
                d0 = time.perf_counter()
                for _ in range(10):
                    e.decide(mask)
                e.sync()
                full_cycle["decide_ms"] = (time.perf_counter() - d0) * 1e3 / 10
                full_cycle["decide_what"] = "sweep + per-row argmax in one pass, no score table written; same decisions as eval_argmax"
            except Exception as ex:
                full_cycle["decide_error"] = repr(ex)[:200]
            # the same pods scheduled strictly one after the other, each seeing the commits before it (upstream's
            # semantics; inherently sequential, one workgroup): spx_commit_sequential
            c4 = time.perf_counter()
            seq_node, _, _, _ = e.commit_sequential(mask, want_ties=False)
            c5 = time.perf_counter()
            full_cycle["sequential_commit_ms"] = (c5 - c4) * 1e3
            full_cycle["sequential_pods_per_s"] = n_pods / (c5 - c4)
            full_cycle["sequential_distinct_nodes"] = int(len(set(seq_node.tolist())))
        except Exception as ex:
            full_cycle = {"error": repr(ex)[:200]}

    # the exchange step of the sharded path, reported separately: per-pod decisions, optionally one table
    gather_info = None
    if dist is not None and args.gather != "none":
        try:
            from scheduler_plugins_amd import shard
            score_mask = mask & ~(1 << 6)
            e.eval_best(score_mask)
            node, score, ties, feas = e.best()
            barrier()
            t1 = time.perf_counter()
            shard.gather_best(dist, torch.device("cuda", local_rank), node, score, ties, feas, n_pods * world)
            barrier()
            gather_info = {"best_ms": (time.perf_counter() - t1) * 1e3, "bytes_per_rank": int(n_pods) * 32}
            if args.gather == "table":
                p0 = plugins[-1] if plugins[-1] <= 4 else plugins[0]
                ptr, stride, rows = e.score_table(p0)
                slab = torch.empty((rows, stride), dtype=torch.uint8, device=f"cuda:{local_rank}")
                e.bind_score_table(p0, slab.data_ptr(), stride, rows)
                e.eval(1 << p0)
                e.sync()
                barrier()
                t1 = time.perf_counter()
                full = shard.gather_table(dist, slab)
                barrier()
                gather_info.update({"table_ms": (time.perf_counter() - t1) * 1e3, "table_bytes": int(full.numel())})
                del full
        except Exception as ex:  # never lose the bench line to the optional exchange measurement
            gather_info = {"error": repr(ex)[:200]}

    evals_per_step = n_nodes * n_pods * world
    value = evals_per_step * args.steps / elapsed
    algo_bytes = n_nodes * w["node_row"] + n_pods * w["pod_row"] + n_nodes * n_pods * w["out"]
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9

    traffic, traffic_src = measured_traffic(args.workload) if not args.plugins else (None, None)
    out = {
        "metric": "pod_x_node_filter_score_evals_per_sec",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        # the arithmetic the sweep computes in (results are bit-exact against the reference's float64 / int64 either way):
        # TLP/LVRB float32 with a per-cell exactness proof and a float64 fallback, NRT float64 (exact integers), NetworkOverhead int32
        "dtype": {"nrt": "f64", "net": "i32", "cap": "f32+f64+i32", "lroc": "f64", "peaks": "f64"}.get(w["plugins"][0], "f32+f64"),
        "data": "synthetic",
        "config": {"workload": w["desc"], "n_nodes": n_nodes, "n_pods_per_gpu": n_pods, "plugins": list(w["plugins"]),
                   "sharding": "pod rows per rank, node tables replicated, no data-path collective",
                   "result_tables": "uint8 [pods][nodes] per plugin, resident in HBM"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": {"nrt": "spx::k_nrt_fast (LeastAllocated: Filter launch + Score launch, both counted)", "net": "spx::k_net_cls", "lroc": "spx::k_lroc_fast (float32 quotient on exact float64 numerator/denominator, float64 fallback; VALU-bound)",
                                "peaks": "spx::k_peaks<min/max pass> + spx::k_peaks<write pass> (VALU-bound: division + exp per cell and pass)",
                                "cap": "full profile: k_quota, k_nrt_fast x2, k_net_cls, k_tlp_fast2, k_lvrb_fast, k_alloc_masked"}.get(
                         w["plugins"][0], "spx::k_tlp_fast2 (Allocatable+TLP)" + (" + spx::k_lvrb_fast" if "lvrb" in w["plugins"] else "")),
                     "kernel_ms": kern_ms,
                     "algorithmic_bytes": algo_bytes, "frac_of_measured_copy_ceiling_6.29TBs": achieved / 6290.0},
        "kernel_evals_per_sec": n_nodes * n_pods / (kern_ms * 1e-3),
    }
    if full_cycle is not None:
        out["full_cycle"] = full_cycle
    if gather_info is not None:
        out["gather"] = gather_info
    if rank == 0 and world == 1 and args.cpu_budget > 0:
        out["cpu_baseline"] = cpu_baseline(spx, snap, e, plugins, args.cpu_budget)
    e.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
