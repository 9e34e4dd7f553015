#!/usr/bin/env python3
"""Rewrites the method bodies of the reference's plugins to read libspx's rows — the patch a maintainer would apply, carried as
"replace the body of method M in file F with B" so that no reference source is stored here.

usage: apply_shim.py <checkout of kubernetes-sigs/scheduler-plugins> <output dir>
Writes the edited copies of the six files under <output dir> (same relative paths) and prints a summary.  Besides the bodies a
maintainer adds: the field `spx *spx.Engine` to Allocatable / TargetLoadPacking / TopologyMatch / NetworkOverhead (set in their
New functions from the profile's engine), `order map[string]int32` to TopologicalSort, the import of pkg/spx (shim/go/pkg/spx),
and renames the reference's TopologyMatch.Filter body to filterWithVictims for the preemption dry run.  UNCOMPILED: no Go
toolchain in this image; tests/test_go_shim_apply.py checks that every edit still finds its method in /root/reference.

Reference methods replaced (file:line in the surveyed checkout):
  pkg/noderesources/allocatable.go:63 Score, :143 NormalizeScore
  pkg/trimaran/targetloadpacking/targetloadpacking.go:107 Score
  pkg/noderesourcetopology/filter.go:179 Filter, score.go:62 Score
  pkg/networkaware/networkoverhead/networkoverhead.go:326 Filter, :362 Score, :389 NormalizeScore
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
	row, err := no.spx.RawRow(pod, spx.PluginNetOverhead, 0) // the accumulated cost (getAccumulatedCost :576-638)
	if err != nil {
		return 0, fwk.AsStatus(err)
	}
	return row[no.spx.Column(nodeName)], nil"""),
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
