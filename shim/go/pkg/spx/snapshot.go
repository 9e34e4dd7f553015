package spx

/*
#include <stdlib.h>
#include <string.h>
#include "spx.h"
*/
import "C"

import "unsafe"

// cArray copies a Go slice into C memory (no Go pointer may be retained by, or nested in, what crosses the boundary) and returns
// the C pointer plus its release function.
func cArray[T any](s []T) (unsafe.Pointer, func()) {
	if len(s) == 0 {
		return nil, func() {}
	}
	n := C.size_t(len(s)) * C.size_t(unsafe.Sizeof(s[0]))
	p := C.malloc(n)
	C.memcpy(p, unsafe.Pointer(unsafe.SliceData(s)), n)
	return p, func() { C.free(p) }
}

// TrimaranNodes are the columns of spx_trimaran_nodes_soa for a set of nodes: the flat form of what Collector.GetNodeMetrics,
// node.Status.Capacity / Allocatable and the ScheduledPodsCache hold (collector.go:110-123, resourcestats.go:45-107,
// handler.go:47-58).  The shim fills them from its informer caches (or hands the raw objects to spx_flatten_trimaran_nodes).
type TrimaranNodes struct {
	CapCPUMilli, MissingMilli, AllocCPUMilli, AllocMem []int64
	CPUUtil, CPUAvg, CPUStd, MemAvg, MemStd            []float64
	TLPValid, LVFlags                                  []uint8
}

func (t *TrimaranNodes) soa() (C.spx_trimaran_nodes_soa, func()) {
	var frees []func()
	p := func(ptr unsafe.Pointer, f func()) unsafe.Pointer { frees = append(frees, f); return ptr }
	s := C.spx_trimaran_nodes_soa{n_nodes: C.int64_t(len(t.CapCPUMilli))}
	s.cap_cpu_milli = (*C.int64_t)(p(cArray(t.CapCPUMilli)))
	s.tlp_cpu_util = (*C.double)(p(cArray(t.CPUUtil)))
	s.tlp_missing_milli = (*C.int64_t)(p(cArray(t.MissingMilli)))
	s.tlp_valid = (*C.uint8_t)(p(cArray(t.TLPValid)))
	s.lv_alloc_cpu_milli = (*C.int64_t)(p(cArray(t.AllocCPUMilli)))
	s.lv_alloc_mem = (*C.int64_t)(p(cArray(t.AllocMem)))
	s.lv_cpu_avg = (*C.double)(p(cArray(t.CPUAvg)))
	s.lv_cpu_std = (*C.double)(p(cArray(t.CPUStd)))
	s.lv_mem_avg = (*C.double)(p(cArray(t.MemAvg)))
	s.lv_mem_std = (*C.double)(p(cArray(t.MemStd)))
	s.lv_flags = (*C.uint8_t)(p(cArray(t.LVFlags)))
	return s, func() {
		for _, f := range frees {
			f()
		}
	}
}

// UploadTrimaranNodes replaces the whole table (once per snapshot).
func (e *Engine) UploadTrimaranNodes(t *TrimaranNodes) error {
	s, free := t.soa()
	defer free()
	if rc := C.spx_upload_trimaran_nodes(e.h, &s); rc != 0 {
		return e.err("spx_upload_trimaran_nodes")
	}
	return nil
}

// UpdateTrimaranNodes replaces the rows of the nodes that changed since the last cycle (a collector refresh, a bind that entered
// the ScheduledPodsCache): rows[i] describes node columns[i].  Staged as one blob and scattered on the device.
func (e *Engine) UpdateTrimaranNodes(columns []int64, rows *TrimaranNodes) error {
	s, free := rows.soa()
	defer free()
	idx, freeIdx := cArray(columns)
	defer freeIdx()
	if rc := C.spx_update_trimaran_nodes(e.h, (*C.int64_t)(idx), &s); rc != 0 {
		return e.err("spx_update_trimaran_nodes")
	}
	return nil
}

// UploadTrimaranPods uploads the pending batch's pod columns: predicted CPU (PredictUtilisation, targetloadpacking.go:198-205)
// and the LVRB requests (resourcestats.go:110-146), one entry per pod row.
func (e *Engine) UploadTrimaranPods(tlpMilli, reqCPUMilli, reqMem []int64) error {
	a, fa := cArray(tlpMilli)
	defer fa()
	b, fb := cArray(reqCPUMilli)
	defer fb()
	c, fc := cArray(reqMem)
	defer fc()
	s := C.spx_trimaran_pods_soa{n_pods: C.int64_t(len(tlpMilli)), tlp_pod_milli: (*C.int64_t)(a), lv_req_cpu_milli: (*C.int64_t)(b), lv_req_mem: (*C.int64_t)(c)}
	if rc := C.spx_upload_trimaran_pods(e.h, &s); rc != 0 {
		return e.err("spx_upload_trimaran_pods")
	}
	return nil
}

// IngestNRT hands the JSON of a NodeResourceTopology list (or of the objects of a watch batch) to the library instead of
// marshalling Go structs field by field (pluginhelpers.go:105-161, nodeconfig/topologymanager.go:78-162); the object table it
// returns feeds spx_flatten_nrt_nodes / spx_flatten_nrt_node_rows + spx_upload_nrt_nodes / spx_update_nrt_nodes.
func IngestNRT(h *C.spx_ingest, doc []byte) (objects, unknown int64, err error) {
	p := C.CBytes(doc)
	defer C.free(p)
	var n, u C.int64_t
	if rc := C.spx_ingest_nrt_json(h, (*C.char)(p), C.int64_t(len(doc)), &n, &u); rc != 0 {
		return 0, 0, fmtIngestError(h)
	}
	return int64(n), int64(u), nil
}
