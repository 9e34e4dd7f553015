package spx

/*
#include <stdlib.h>
#include "spx.h"
*/
import "C"

import (
	"fmt"
	"unsafe"

	v1 "k8s.io/api/core/v1"
)

// Decisions is what upstream's selectHost needs per pending pod (row order = the batch's pod order): the best node's column
// (-1: no feasible node / rejected by a PreFilter), the weighted score sum it reached, how many nodes tie at that sum (upstream
// draws one of them; Node is the lowest column) and how many nodes passed every Filter.
type Decisions struct {
	Node     []int32
	Score    []int64
	Ties     []int32
	Feasible []int32
}

func (e *Engine) fetchBest(n int64) (*Decisions, error) {
	d := &Decisions{Node: make([]int32, n), Score: make([]int64, n), Ties: make([]int32, n), Feasible: make([]int32, n)}
	if n == 0 {
		return d, nil
	}
	if rc := C.spx_fetch_best(e.h, 0, C.int64_t(n), (*C.int32_t)(unsafe.Pointer(&d.Node[0])), (*C.int64_t)(unsafe.Pointer(&d.Score[0])),
		(*C.int32_t)(unsafe.Pointer(&d.Ties[0])), (*C.int32_t)(unsafe.Pointer(&d.Feasible[0]))); rc != 0 {
		return nil, e.err("spx_fetch_best")
	}
	return d, nil
}

// Decide evaluates the profile `mask` for the whole pending batch on the frozen snapshot and returns the per-pod decisions
// without materialising score tables where the profile allows it (spx_decide): one sweep, 20 bytes per pod back.
func (e *Engine) Decide(mask uint32, nPods int64) (*Decisions, error) {
	e.mu.Lock() // drives the engine stream like Eval: one writer at a time
	defer e.mu.Unlock()
	if rc := C.spx_decide(e.h, C.uint32_t(mask), 0, C.int64_t(nPods)); rc != 0 {
		return nil, e.err("spx_decide")
	}
	d, err := e.fetchBest(nPods)
	// spx_decide may (re)write the Filter plugins' tables: rows cached before the call are not served after it
	old := e.gen.Load()
	e.gen.Store(&generation{nNodes: old.nNodes, nPods: old.nPods, column: old.column, podRow: old.podRow})
	return d, err
}

// CommitSequential schedules the batch one pod after the other on the device, every pod seeing the commits of the pods before
// it (what the plugins' Reserve hooks and event handlers do between two scheduling cycles: overreserve.go:170-203,
// elasticquota.go:89-97, handler.go:131-139, the AppGroup's scheduled list) — spx_commit_sequential.  The Go side still performs
// the real Reserve / Bind from the returned node per pod.
func (e *Engine) CommitSequential(mask uint32, nPods int64) (*Decisions, error) {
	d := &Decisions{Node: make([]int32, nPods), Score: make([]int64, nPods), Ties: make([]int32, nPods)}
	if nPods == 0 {
		return d, nil
	}
	e.mu.Lock()
	defer e.mu.Unlock()
	if rc := C.spx_commit_sequential(e.h, C.uint32_t(mask), 0, C.int64_t(nPods), (*C.int32_t)(unsafe.Pointer(&d.Node[0])),
		(*C.int64_t)(unsafe.Pointer(&d.Score[0])), (*C.int32_t)(unsafe.Pointer(&d.Ties[0])), nil); rc != 0 {
		return nil, e.err("spx_commit_sequential")
	}
	// the loop evaluates single rows into the plugins' tables while it runs (and restores the snapshot's tables when it ends): a fresh
	// generation, so that no row fetched before or during the call is served after it
	old := e.gen.Load()
	e.gen.Store(&generation{nNodes: old.nNodes, nPods: old.nPods, column: old.column, podRow: old.podRow})
	return d, nil
}

// LoadTrimaranPods replaces the pending batch of the trimaran / Allocatable tables (the node tables stay): a cycle's new pods.
// Rows cached for the previous batch are dropped with the generation they belong to.
func (e *Engine) LoadTrimaranPods(in *Ingest, pods []*v1.Pod) error {
	e.mu.Lock()
	defer e.mu.Unlock()
	if rc := C.spx_load_trimaran_pods(e.h, C.spx_ingest_pod_objects(in.h)); rc != 0 {
		return e.err("spx_load_trimaran_pods")
	}
	old := e.gen.Load()
	g := &generation{nNodes: old.nNodes, nPods: int64(len(pods)), column: old.column, podRow: make(map[string]int64, len(pods))}
	for i, p := range pods {
		g.podRow[string(p.UID)] = int64(i)
	}
	e.gen.Store(g) // fresh row / raw caches: nothing fetched for the old batch is served for the new one
	return nil
}

// PlacedPod is a pod that joined an AppGroup's scheduled list since the tables were loaded (bound by an earlier cycle or by
// another scheduler): its AppGroup and workload selector as the ingest handle numbers them, and its node's column.
type PlacedPod struct {
	Group, Selector, Node int32
}

// UpdateNetPlaced appends what those pods add to NetworkOverhead's (host, MaxNetworkCost) pair lists on the device
// (spx_flatten_net_placed + spx_update_net_placed; networkoverhead.go:654-694) — equal to reloading the grown AppGroups.
func (e *Engine) UpdateNetPlaced(in *Ingest, placed []PlacedPod) error {
	n := len(placed)
	if n == 0 {
		return nil
	}
	group, selector, node := make([]int32, n), make([]int32, n), make([]int32, n)
	for i, p := range placed {
		group[i], selector[i], node[i] = p.Group, p.Selector, p.Node
	}
	gp, sp, np := (*C.int32_t)(unsafe.Pointer(&group[0])), (*C.int32_t)(unsafe.Pointer(&selector[0])), (*C.int32_t)(unsafe.Pointer(&node[0]))
	var entries C.int64_t
	// (the flatteners are engine-less: their failures carry no engine message — e.err would hand back a stale one)
	if rc := C.spx_flatten_net_placed(C.spx_ingest_pod_objects(in.h), C.spx_ingest_appgroup_objects(in.h), C.int64_t(n), gp, sp, np, &entries, nil, nil, nil); rc != 0 {
		return fmt.Errorf("spx_flatten_net_placed: rc %d (group / selector / node index out of range?)", int(rc))
	}
	if entries == 0 {
		return nil
	}
	key, at, cost := make([]int32, entries), make([]int32, entries), make([]int64, entries)
	if rc := C.spx_flatten_net_placed(C.spx_ingest_pod_objects(in.h), C.spx_ingest_appgroup_objects(in.h), C.int64_t(n), gp, sp, np, &entries,
		(*C.int32_t)(unsafe.Pointer(&key[0])), (*C.int32_t)(unsafe.Pointer(&at[0])), (*C.int64_t)(unsafe.Pointer(&cost[0]))); rc != 0 {
		return fmt.Errorf("spx_flatten_net_placed: rc %d", int(rc))
	}
	e.mu.Lock()
	defer e.mu.Unlock()
	if rc := C.spx_update_net_placed(e.h, entries, (*C.int32_t)(unsafe.Pointer(&key[0])), (*C.int32_t)(unsafe.Pointer(&at[0])),
		(*C.int64_t)(unsafe.Pointer(&cost[0]))); rc != 0 {
		return e.err("spx_update_net_placed")
	}
	old := e.gen.Load() // NetworkOverhead's rows changed: readers get a generation with empty row caches
	e.gen.Store(&generation{nNodes: old.nNodes, nPods: old.nPods, column: old.column, podRow: old.podRow})
	return nil
}
