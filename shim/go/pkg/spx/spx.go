// Package spx binds libspx.so (include/spx.h) for the reference's Go plugins: the cgo shim north_star names.
//
// UNCOMPILED: this image has no Go toolchain (SURVEY.md 8c).  The file is source for review — every C entry point it calls is
// exercised with the same call pattern by the C++ host mirror (tests/cpp/harness.cc: 16 concurrent readers per pod) and by the
// ctypes binding (scheduler-plugins_amd/engine.py) against the identical ABI; tests/test_go_shim_apply.py tokenises these files
// and the method bodies of apply_shim.py and checks every identifier they use against what is declared (parameters, receivers,
// locals, package-level names, imports) — the class of defect a compiler would have caught in round 3.
//
// cgo rules the ABI was shaped for: every argument is a pointer to a flat array of fixed-width scalars or a struct of such
// pointers; nothing is retained after a call returns; no callbacks into Go; row fetches are read-only after Sync and may run on
// many goroutines at once (each OS thread stages through its own pinned buffer inside the library).
package spx

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../../scheduler-plugins_amd -lspx
#include <stdlib.h>
#include "spx.h"
*/
import "C"

import (
	"fmt"
	"sync"
	"sync/atomic"
	"unsafe"

	v1 "k8s.io/api/core/v1"
	fwk "k8s.io/kube-scheduler/framework"
)

// Plugin ids (SPX_PLUGIN_*).
const (
	PluginAllocatable = int(C.SPX_PLUGIN_ALLOCATABLE)
	PluginTLP         = int(C.SPX_PLUGIN_TLP)
	PluginLVRB        = int(C.SPX_PLUGIN_LVRB)
	PluginNRT         = int(C.SPX_PLUGIN_NRT)
	PluginNetOverhead = int(C.SPX_PLUGIN_NETOVERHEAD)
	PluginCapacity    = int(C.SPX_PLUGIN_CAPACITY)
)

// CapacityScheduling.PreFilter verdicts (SPX_QUOTA_ST_*).
const (
	QuotaOverMax = byte(C.SPX_QUOTA_ST_OVER_MAX)
	QuotaOverMin = byte(C.SPX_QUOTA_ST_OVER_MIN)
)

// rowBlock rows are fetched per call: one row per call runs at 0.27 M rows/s under the reference's fan-out of 16 readers, a
// 64-row block at 3.0 M rows/s (10k-node rows, tests/cpp/harness.cc boundary_throughput) — the queue is evaluated in order,
// so the pods after the current one are the ones asked for next.
const rowBlock = 64

// Engine is one scheduler profile on one GPU.  No package-level state (unlike targetloadpacking.go:49-53).
type Engine struct {
	h   *C.spx_engine
	mu  sync.Mutex                 // serialises Bind / Eval (writers); readers never take it
	gen atomic.Pointer[generation] // what readers see: swapped as a whole, never mutated in place
}

// generation is everything a row reader needs, immutable once published: the batch's shape, the name -> column / UID -> row maps
// and the caches of fetched rows.  Bind and Eval publish a fresh one (a sync.Map must not be copied or reassigned while readers
// hold it; swapping the pointer leaves late readers of the old generation with a consistent, if stale, view).
type generation struct {
	nNodes int
	nPods  int64
	column map[string]int32 // node name -> column of the snapshot
	podRow map[string]int64 // pod UID -> row of the evaluated batch
	blocks sync.Map         // blockKey -> *block: score / status rows fetched so far
	raws   sync.Map         // rawKey -> *rawRow: int64 rows fetched so far
}

type blockKey struct {
	plugin, kind int // kind 0 scores, 1 status
	first        int64
}

type block struct {
	once sync.Once
	rows []byte // [rows][nNodes]
	n    int64
	err  error
}

type rawKey struct {
	plugin, which int
	row           int64
}

type rawRow struct {
	once sync.Once
	v    []int64
	err  error
}

// New creates the engine on HIP device `device`; there is no CPU fallback (SPX_ERR_NOGPU).
func New(device int) (*Engine, error) {
	var h *C.spx_engine
	if rc := C.spx_create(C.int(device), &h); rc != 0 {
		return nil, fmt.Errorf("spx_create: %s", C.GoString(C.spx_last_error(nil)))
	}
	e := &Engine{h: h}
	e.gen.Store(&generation{column: map[string]int32{}, podRow: map[string]int64{}})
	return e, nil
}

func (e *Engine) Close() { C.spx_destroy(e.h) }

func (e *Engine) err(what string) error {
	return fmt.Errorf("%s: %s", what, C.GoString(C.spx_last_error(e.h))) // the calling thread's own message
}

// Bind records which column a node and which row a pending pod occupy in the tables the caller is about to upload.
func (e *Engine) Bind(nodes []string, pods []*v1.Pod) {
	g := &generation{nNodes: len(nodes), nPods: int64(len(pods)), column: make(map[string]int32, len(nodes)), podRow: make(map[string]int64, len(pods))}
	for i, n := range nodes {
		g.column[n] = int32(i)
	}
	for i, p := range pods {
		g.podRow[string(p.UID)] = int64(i)
	}
	e.mu.Lock()
	e.gen.Store(g)
	e.mu.Unlock()
}

// Column is the snapshot column of a node (-1: not in the snapshot).
func (e *Engine) Column(node string) int32 {
	if c, ok := e.gen.Load().column[node]; ok {
		return c
	}
	return -1
}

// Eval runs the batched sweep of the plugins in mask over every pending pod and waits for it; rows are readable afterwards.
func (e *Engine) Eval(mask uint32) error {
	e.mu.Lock()
	defer e.mu.Unlock()
	old := e.gen.Load()
	if rc := C.spx_eval(e.h, C.uint32_t(mask), 0, C.int64_t(old.nPods)); rc != 0 {
		return e.err("spx_eval")
	}
	if rc := C.spx_sync(e.h); rc != 0 {
		return e.err("spx_sync")
	}
	// same shape and maps, empty row caches
	e.gen.Store(&generation{nNodes: old.nNodes, nPods: old.nPods, column: old.column, podRow: old.podRow})
	return nil
}

// copyOut copies n bytes of C memory into a Go slice (C.GoBytes takes a C.int: a block of 64 rows x 40M nodes would overflow it)
func copyOut(p unsafe.Pointer, n int64) []byte {
	out := make([]byte, n)
	copy(out, unsafe.Slice((*byte)(p), n))
	return out
}

func (e *Engine) row(pod *v1.Pod, plugin, kind int) ([]byte, error) {
	g := e.gen.Load()
	r, ok := g.podRow[string(pod.UID)]
	if !ok {
		return nil, fmt.Errorf("pod %s/%s is not in the evaluated batch", pod.Namespace, pod.Name)
	}
	first := r - r%rowBlock
	v, _ := g.blocks.LoadOrStore(blockKey{plugin, kind, first}, &block{})
	b := v.(*block)
	b.once.Do(func() { // the first of the 16 Parallelizer goroutines fetches, the others wait on the Once
		b.n = g.nPods - first
		if b.n > rowBlock {
			b.n = rowBlock
		}
		size := b.n * int64(g.nNodes)
		buf := C.malloc(C.size_t(size)) // C memory: no Go pointer crosses the boundary
		defer C.free(buf)
		var rc C.int
		if kind == 0 {
			rc = C.spx_fetch_score_rows(e.h, C.int(plugin), C.int64_t(first), C.int64_t(first+b.n), (*C.uint8_t)(buf), C.int64_t(g.nNodes))
		} else {
			rc = C.spx_fetch_status_rows(e.h, C.int(plugin), C.int64_t(first), C.int64_t(first+b.n), (*C.uint8_t)(buf), C.int64_t(g.nNodes))
		}
		if rc != 0 {
			b.err = e.err("spx_fetch_rows")
			return
		}
		b.rows = copyOut(buf, size)
	})
	if b.err != nil {
		return nil, b.err
	}
	off := (r - first) * int64(g.nNodes)
	return b.rows[off : off+int64(g.nNodes)], nil
}

// ScoreRow is the pod's normalised 0..100 row of a Score plugin (what NormalizeScore leaves); StatusRow a Filter plugin's codes.
func (e *Engine) ScoreRow(pod *v1.Pod, plugin int) ([]byte, error)  { return e.row(pod, plugin, 0) }
func (e *Engine) StatusRow(pod *v1.Pod, plugin int) ([]byte, error) { return e.row(pod, plugin, 1) }

// RawRow is the int64 row a direct caller of Score() observes before NormalizeScore (Allocatable's negative sums,
// NetworkOverhead's accumulated cost: `which` 0 cost / 1 satisfied / 2 violated).  Computed on demand ONCE per (pod, plugin,
// which) — Score() asks for it per node, 10 000 times per pod — and cached for the generation; concurrent first callers wait on
// the Once, concurrent launches are serialised inside the library.
func (e *Engine) RawRow(pod *v1.Pod, plugin, which int) ([]int64, error) {
	g := e.gen.Load()
	r, ok := g.podRow[string(pod.UID)]
	if !ok {
		return nil, fmt.Errorf("pod %s/%s is not in the evaluated batch", pod.Namespace, pod.Name)
	}
	v, _ := g.raws.LoadOrStore(rawKey{plugin, which, r}, &rawRow{})
	rr := v.(*rawRow)
	rr.once.Do(func() {
		buf := C.malloc(C.size_t(g.nNodes) * 8)
		defer C.free(buf)
		if rc := C.spx_fetch_raw(e.h, C.int(plugin), C.int(which), C.int64_t(r), (*C.int64_t)(buf)); rc != 0 {
			rr.err = e.err("spx_fetch_raw")
			return
		}
		rr.v = make([]int64, g.nNodes)
		copy(rr.v, unsafe.Slice((*int64)(buf), g.nNodes))
	})
	return rr.v, rr.err
}

// PreFilter is CapacityScheduling.PreFilter's verdict for the pod: 0, SPX_QUOTA_ST_OVER_MAX or SPX_QUOTA_ST_OVER_MIN.
func (e *Engine) PreFilter(pod *v1.Pod) (byte, error) {
	r, ok := e.gen.Load().podRow[string(pod.UID)]
	if !ok {
		return 0, fmt.Errorf("pod %s/%s is not in the evaluated batch", pod.Namespace, pod.Name)
	}
	var st C.uint8_t
	if rc := C.spx_fetch_prefilter(e.h, C.SPX_PLUGIN_CAPACITY, C.int64_t(r), C.int64_t(r+1), &st); rc != 0 {
		return 0, e.err("spx_fetch_prefilter")
	}
	return byte(st), nil
}

// NRTStatus turns a TopologyMatch.Filter code into the reference's status (filter.go:42-245).
func NRTStatus(code byte) *fwk.Status {
	switch code {
	case 0:
		return nil
	case C.SPX_NRT_ST_INVALID_TOPOLOGY:
		return fwk.NewStatus(fwk.Unschedulable, "invalid node topology data")
	case C.SPX_NRT_ST_INIT_CONTAINER:
		return fwk.NewStatus(fwk.Unschedulable, "cannot align init container")
	case C.SPX_NRT_ST_SIDECAR_CONTAINER:
		return fwk.NewStatus(fwk.Unschedulable, "cannot align sidecar container")
	case C.SPX_NRT_ST_CONTAINER:
		return fwk.NewStatus(fwk.Unschedulable, "cannot align container")
	case C.SPX_NRT_ST_POD:
		return fwk.NewStatus(fwk.Unschedulable, "cannot align pod")
	}
	return fwk.NewStatus(fwk.Error, fmt.Sprintf("unknown NRT status %d", code))
}
