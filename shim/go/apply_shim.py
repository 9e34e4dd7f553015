#!/usr/bin/env python3
"""Rewrites the method bodies of the reference's plugins to read libspx's rows — the patch a maintainer would apply, carried as
"replace the body of method M in file F with B" so that no reference source is stored here.

usage: apply_shim.py <checkout of kubernetes-sigs/scheduler-plugins> <output dir>
Writes the edited copies of the six files under <output dir> (same relative paths) and prints a summary.  Besides the bodies a
maintainer adds: the field `spx *spx.Engine` to Allocatable / TargetLoadPacking / LoadVariationRiskBalancing / TopologyMatch /
NetworkOverhead / CapacityScheduling (set in their New functions from the profile's engine), `order map[string]int32` to TopologicalSort, the import of pkg/spx (shim/go/pkg/spx),
and renames the reference's TopologyMatch.Filter body to filterWithVictims for the preemption dry run.  UNCOMPILED: no Go
toolchain in this image; tests/test_go_shim_apply.py checks that every edit still finds its method in /root/reference.

Reference methods replaced (file:line in the surveyed checkout; twelve bodies in eight files):
  pkg/noderesources/allocatable.go:63 Score, :143 NormalizeScore
  pkg/trimaran/targetloadpacking/targetloadpacking.go:107 Score
  pkg/trimaran/loadvariationriskbalancing/loadvariationriskbalancing.go:84 Score
  pkg/capacityscheduling/capacity_scheduling.go:208 PreFilter
  pkg/noderesourcetopology/filter.go:179 Filter, score.go:62 Score
  pkg/networkaware/networkoverhead/networkoverhead.go:174 PreFilter, :326 Filter, :362 Score, :389 NormalizeScore
  pkg/networkaware/topologicalsort/topologicalsort.go:102 Less
"""
import re
import sys
from pathlib import Path

EDITS = {
    "pkg/noderesources/allocatable.go": [
        (r"^func \(alloc \*Allocatable\) Score\([^\n]*\{", """\
	// the row libspx computed for this pod over every node of the snapshot (spx_fetch_raw: the int64 a direct caller of Score
	// observes); alloc.spx is the profile's *spx.Engine, set in NewAllocatable
	row, err := alloc.spx.RawRow(pod, spx.PluginAllocatable, 0)
	if err != nil {
		return 0, fwk.AsStatus(err)
	}
	return row[alloc.spx.Column(nodeInfo.Node().Name)], nil"""),
        (r"^func \(alloc \*Allocatable\) NormalizeScore\([^\n]*\{", """\
	// rows arrive normalised over the pod's feasible nodes (the engine's own Filter tables, or the mask the shim uploaded with
	// spx_upload_feasible_mask when Filter plugins outside the engine took part)
	row, err := alloc.spx.ScoreRow(pod, spx.PluginAllocatable)
	if err != nil {
		return fwk.AsStatus(err)
	}
	for i := range scores {
		scores[i].Score = int64(row[alloc.spx.Column(scores[i].Name)])
	}
	return nil"""),
    ],
    "pkg/trimaran/targetloadpacking/targetloadpacking.go": [
        (r"^func \(pl \*TargetLoadPacking\) Score\([^\n]*\{", """\
	row, err := pl.spx.ScoreRow(pod, spx.PluginTLP)
	if err != nil {
		return fwk.MinNodeScore, fwk.AsStatus(err)
	}
	return int64(row[pl.spx.Column(nodeInfo.Node().Name)]), nil"""),
    ],
    "pkg/trimaran/loadvariationriskbalancing/loadvariationriskbalancing.go": [
        (r"^func \(pl \*LoadVariationRiskBalancing\) Score\([^\n]*\{", """\
	row, err := pl.spx.ScoreRow(pod, spx.PluginLVRB)
	if err != nil {
		return fwk.MinNodeScore, fwk.AsStatus(err)
	}
	return int64(row[pl.spx.Column(nodeInfo.Node().Name)]), nil"""),
    ],
    "pkg/capacityscheduling/capacity_scheduling.go": [
        (r"^func \(c \*CapacityScheduling\) PreFilter\([^\n]*\{", """\
	// the snapshot and the preFilterState stay (PostFilter's preemption reads them); the two cmp2 gates and the nominated-pod
	// walk (:231-283) were evaluated for the whole pending batch by spx_eval (k_quota)
	snapshotElasticQuota := c.snapshotElasticQuota()
	state.Write(ElasticQuotaSnapshotKey, snapshotElasticQuota)
	podReq := computePodResourceRequest(pod)
	state.Write(preFilterStateKey, &PreFilterState{podReq: *podReq})
	verdict, err := c.spx.PreFilter(pod)
	if err != nil {
		return nil, fwk.AsStatus(err)
	}
	switch verdict {
	case spx.QuotaOverMax:
		return nil, fwk.NewStatus(fwk.Unschedulable, fmt.Sprintf("Pod %v/%v is rejected in PreFilter because ElasticQuota %v is more than Max", pod.Namespace, pod.Name, pod.Namespace))
	case spx.QuotaOverMin:
		return nil, fwk.NewStatus(fwk.Unschedulable, fmt.Sprintf("Pod %v/%v is rejected in PreFilter because total ElasticQuota used is more than min", pod.Namespace, pod.Name))
	}
	return nil, fwk.NewStatus(fwk.Success, "")"""),
    ],
    "pkg/noderesourcetopology/filter.go": [
        (r"^func \(tm \*TopologyMatch\) Filter\([^\n]*\{", """\
	if nodeInfo.Node() == nil {
		return fwk.NewStatus(fwk.Error, "node not found")
	}
	nodeName := nodeInfo.Node().Name
	if victims, _ := getVictimPods(cycleState, tm.preemptionMode); len(victims) > 0 {
		// preemption dry run: the candidate's post-eviction zone table is another snapshot row (spx_nrt_post_eviction +
		// spx_update_nrt_nodes on a scratch engine); the reference path, kept under this name, serves it
		return tm.filterWithVictims(ctx, cycleState, pod, nodeInfo)
	}
	row, err := tm.spx.StatusRow(pod, spx.PluginNRT)
	if err != nil {
		return fwk.AsStatus(err)
	}
	code := row[tm.spx.Column(nodeName)]
	status := spx.NRTStatus(code)
	if code > 1 { // an alignment failure, not stale data (SPX_NRT_ST_INVALID_TOPOLOGY == 1): the bookkeeping of filter.go:241-243
		tm.nrtCache.NodeMaybeOverReserved(nodeName, pod)
	}
	return status"""),
    ],
    "pkg/noderesourcetopology/score.go": [
        (r"^func \(tm \*TopologyMatch\) Score\([^\n]*\{", """\
	row, err := tm.spx.ScoreRow(pod, spx.PluginNRT)
	if err != nil {
		return 0, fwk.AsStatus(err)
	}
	return int64(row[tm.spx.Column(nodeInfo.Node().Name)]), nil"""),
    ],
    "pkg/networkaware/networkoverhead/networkoverhead.go": [
        (r"^func \(no \*NetworkOverhead\) PreFilter\([^\n]*\{", """\
	// the per-pod work of PreFilter — AppGroup / NetworkTopology lookups, the dependency and scheduled lists, the cost map of every
	// node (:174-298) — happened once for the whole batch when it was flattened (spx_flatten_net_keys) and swept (spx_eval).  What the
	// reference decides here per pod is whether Filter and Score have anything to do: a pod without AppGroup, a workload without
	// dependencies or an AppGroup with nothing scheduled yet "scores equally" — the engine's rows carry that (status 0, score 0)
	preFilterState := &PreFilterState{scoreEqually: true}
	state.Write(preFilterStateKey, preFilterState)
	if _, err := no.spx.StatusRow(pod, spx.PluginNetOverhead); err != nil {
		return nil, fwk.AsStatus(err)
	}
	return nil, fwk.NewStatus(fwk.Success, "")"""),
        (r"^func \(no \*NetworkOverhead\) Filter\(ctx context\.Context,\n[^{]*\{", """\
	if nodeInfo.Node() == nil {
		return fwk.NewStatus(fwk.Error, "node not found")
	}
	row, err := no.spx.StatusRow(pod, spx.PluginNetOverhead)
	if err != nil {
		return fwk.AsStatus(err)
	}
	col := no.spx.Column(nodeInfo.Node().Name)
	if row[col] == 0 {
		return nil
	}
	// the message quotes the two counters (networkoverhead.go:352-355): the raw rows hold them
	sat, _ := no.spx.RawRow(pod, spx.PluginNetOverhead, 1)
	vio, _ := no.spx.RawRow(pod, spx.PluginNetOverhead, 2)
	return fwk.NewStatus(fwk.Unschedulable,
		fmt.Sprintf("Node %v does not meet several network requirements from Workload dependencies: Satisfied: %v Violated: %v", nodeInfo.Node().Name, sat[col], vio[col]))"""),
        (r"^func \(no \*NetworkOverhead\) Score\(ctx context\.Context,\n[^{]*\{", """\
	row, err := no.spx.RawRow(pod, spx.PluginNetOverhead, 0) // the accumulated cost (getAccumulatedCost :576-638); cached per pod
	if err != nil {
		return 0, fwk.AsStatus(err)
	}
	return row[no.spx.Column(nodeInfo.Node().Name)], nil"""),
        (r"^func \(no \*NetworkOverhead\) NormalizeScore\(ctx context\.Context,\n[^{]*\{", """\
	row, err := no.spx.ScoreRow(pod, spx.PluginNetOverhead)
	if err != nil {
		return fwk.AsStatus(err)
	}
	for i := range scores {
		scores[i].Score = int64(row[no.spx.Column(scores[i].Name)])
	}
	return nil"""),
    ],
    "pkg/networkaware/topologicalsort/topologicalsort.go": [
        (r"^func \(ts \*TopologicalSort\) Less\(pInfo1, pInfo2 fwk\.QueuedPodInfo\) bool \{", """\
	p1, p2 := pInfo1.GetPodInfo().GetPod(), pInfo2.GetPodInfo().GetPod()
	g1, g2 := networkawareutil.GetPodAppGroupLabel(p1), networkawareutil.GetPodAppGroupLabel(p2)
	if g1 != g2 || len(g1) == 0 {
		s := &queuesort.PrioritySort{}
		return s.Less(pInfo1, pInfo2)
	}
	// FindPodOrder per pod, computed once per pod when the batch was flattened (spx_flatten_net_keys: topo_order) instead of a CR
	// Get + two binary searches per comparison; the queue order itself can come from one device sort (spx_sort_keys)
	return ts.order[string(p1.UID)] <= ts.order[string(p2.UID)]"""),
    ],
}


def replace_body(text: str, signature: str, body: str) -> str:
    m = re.search(signature, text, re.M | re.S)
    if not m:
        raise LookupError(signature)
    start = m.end() - 1
    depth, i = 0, start
    while True:
        c = text[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                break
        i += 1
    return text[:start + 1] + "\n" + body.rstrip("\n") + "\n" + text[i:]


def apply(ref_root: Path, out_root: Path):
    done = []
    for rel, edits in EDITS.items():
        text = (ref_root / rel).read_text()
        for sig, body in edits:
            text = replace_body(text, sig, body)
        dst = out_root / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        dst.write_text(text)
        done.append((rel, len(edits)))
    return done


if __name__ == "__main__":
    for rel, n in apply(Path(sys.argv[1]), Path(sys.argv[2])):
        print(f"{rel}: {n} method bod{'y' if n == 1 else 'ies'} replaced")
