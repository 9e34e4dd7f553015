package spx

/*
#include <stdlib.h>
#include <string.h>
#include "spx.h"
*/
import "C"

import (
	"fmt"
	"unsafe"
)

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

// Ingest decodes the API server's JSON (NodeResourceTopology, v1.Node, v1.Pod, AppGroup, NetworkTopology, ElasticQuota, the
// load-watcher response) into the library's object tables — the alternative to marshalling Go structs field by field
// (pluginhelpers.go:105-161, nodeconfig/topologymanager.go:78-162, networkoverhead.go:448-497, elasticquota.go:48-123).  The
// tables live inside the handle; Load* hands them to an engine.
type Ingest struct {
	h *C.spx_ingest
}

func cStrings(s []string) (**C.char, func()) {
	if len(s) == 0 {
		return nil, func() {}
	}
	arr := (**C.char)(C.malloc(C.size_t(len(s)) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	view := unsafe.Slice(arr, len(s))
	for i, v := range s {
		view[i] = C.CString(v)
	}
	return arr, func() {
		for _, p := range view {
			C.free(unsafe.Pointer(p))
		}
		C.free(unsafe.Pointer(arr))
	}
}

// NewIngest fixes the node order (= snapshot columns) and the resource names the tables will use.
func NewIngest(nodeNames, resourceNames []string) (*Ingest, error) {
	nn, freeN := cStrings(nodeNames)
	defer freeN()
	rn, freeR := cStrings(resourceNames)
	defer freeR()
	var h *C.spx_ingest
	if rc := C.spx_ingest_create(nn, C.int64_t(len(nodeNames)), rn, C.int32_t(len(resourceNames)), &h); rc != 0 {
		return nil, fmt.Errorf("spx_ingest_create failed (%d)", int(rc))
	}
	return &Ingest{h: h}, nil
}

func (in *Ingest) Close() { C.spx_ingest_destroy(in.h) }

type ingestFn func(p *C.char, n C.int64_t) C.int

func (in *Ingest) feed(doc []byte, fn ingestFn) error {
	p := C.CBytes(doc)
	defer C.free(p)
	if rc := fn((*C.char)(p), C.int64_t(len(doc))); rc != 0 {
		return fmtIngestError(in.h)
	}
	return nil
}

// NRT / Nodes / Pods / AppGroups / NetworkTopology / Quotas / Metrics feed one JSON document (a List, or the objects of a watch batch).
func (in *Ingest) NRT(doc []byte) error {
	var n, u C.int64_t
	return in.feed(doc, func(p *C.char, l C.int64_t) C.int { return C.spx_ingest_nrt_json(in.h, p, l, &n, &u) })
}
func (in *Ingest) Nodes(doc []byte) error {
	var n, u C.int64_t
	return in.feed(doc, func(p *C.char, l C.int64_t) C.int { return C.spx_ingest_nodes_json(in.h, p, l, &n, &u) })
}
func (in *Ingest) Pods(doc []byte) error {
	var n C.int64_t
	return in.feed(doc, func(p *C.char, l C.int64_t) C.int { return C.spx_ingest_pods_json(in.h, p, l, &n) })
}
func (in *Ingest) ResetPods() { C.spx_ingest_pods_reset(in.h) }
func (in *Ingest) AppGroups(doc []byte) error {
	var n C.int64_t
	return in.feed(doc, func(p *C.char, l C.int64_t) C.int { return C.spx_ingest_appgroups_json(in.h, p, l, &n) })
}
func (in *Ingest) NetworkTopology(doc []byte, weightsName string) error {
	w := C.CString(weightsName)
	defer C.free(unsafe.Pointer(w))
	return in.feed(doc, func(p *C.char, l C.int64_t) C.int { return C.spx_ingest_nettopo_json(in.h, p, l, w) })
}
func (in *Ingest) Quotas(doc []byte, namespaces []string) error {
	ns, freeNS := cStrings(namespaces)
	defer freeNS()
	var n, u C.int64_t
	return in.feed(doc, func(p *C.char, l C.int64_t) C.int {
		return C.spx_ingest_quota_json(in.h, p, l, ns, C.int32_t(len(namespaces)), &n, &u)
	})
}
func (in *Ingest) Metrics(doc []byte) error {
	var n, u C.int64_t
	return in.feed(doc, func(p *C.char, l C.int64_t) C.int { return C.spx_ingest_metrics_json(in.h, p, l, &n, &u) })
}

// LoadTrimaran / LoadNRT / LoadNetwork / LoadQuota: object tables -> SoA columns -> device in one library call each (spx_load_*):
// the flatteners run with the engine's plugin parameters, nothing is sized or owned on the Go side.
func (e *Engine) LoadTrimaran(in *Ingest) error {
	if rc := C.spx_load_trimaran(e.h, C.spx_ingest_node_objects(in.h), C.spx_ingest_resource_classes(in.h), C.spx_ingest_pod_objects(in.h),
		C.spx_ingest_metrics_objects(in.h), nil); rc != 0 {
		return e.err("spx_load_trimaran")
	}
	return nil
}

// NRTParams is NodeResourceTopologyMatchArgs.ScoringStrategy (apis/config/types.go): Type and the per-resource weights.
type NRTParams struct {
	Strategy  int // SPX_NRT_*
	Resources []string
	Weights   []int64
}

func (e *Engine) LoadNRT(in *Ingest, p NRTParams) error {
	ids := make([]int32, len(p.Resources))
	for i, r := range p.Resources {
		cs := C.CString(r)
		ids[i] = int32(C.spx_ingest_resource_id(in.h, cs))
		C.free(unsafe.Pointer(cs))
	}
	idp, freeID := cArray(ids)
	defer freeID()
	wp, freeW := cArray(p.Weights)
	defer freeW()
	params := C.spx_nrt_params{strategy: C.int32_t(p.Strategy), n_weights: C.int32_t(len(ids)), weight_res: (*C.int32_t)(idp), weight: (*C.int64_t)(wp)}
	if rc := C.spx_load_nrt(e.h, C.spx_ingest_node_objects(in.h), C.spx_ingest_nrt_objects(in.h), C.spx_ingest_resource_classes(in.h),
		C.spx_ingest_pod_objects(in.h), &params); rc != 0 {
		return e.err("spx_load_nrt")
	}
	return nil
}

func (e *Engine) LoadNetwork(in *Ingest) error {
	if rc := C.spx_load_network(e.h, C.spx_ingest_node_objects(in.h), C.spx_ingest_pod_objects(in.h), C.spx_ingest_appgroup_objects(in.h),
		C.spx_ingest_nettopo_objects(in.h)); rc != 0 {
		return e.err("spx_load_network")
	}
	return nil
}

func (e *Engine) LoadQuota(in *Ingest) error {
	if rc := C.spx_load_quota(e.h, C.spx_ingest_pod_objects(in.h), C.spx_ingest_resource_classes(in.h), C.spx_ingest_quota_objects(in.h)); rc != 0 {
		return e.err("spx_load_quota")
	}
	return nil
}

// LoadProfile loads whatever the Ingest holds for the whole profile in ONE library call (spx_load_profile): the four loaders above run
// side by side on host threads of the library (and spx_load_nrt's node and pod halves on two) instead of one after the other from Go —
// 4.5-5 ms against 6.7 for 20 000 nodes x 8 192 pods.  `nrt` == nil skips NodeResourceTopologyMatch; network / quota are skipped when
// `network` / `quota` are false.
func (e *Engine) LoadProfile(in *Ingest, nrt *NRTParams, network, quota bool) error {
	o := C.spx_profile_objects{nodes: C.spx_ingest_node_objects(in.h), rc: C.spx_ingest_resource_classes(in.h), pods: C.spx_ingest_pod_objects(in.h),
		metrics: C.spx_ingest_metrics_objects(in.h)}
	var params C.spx_nrt_params
	if nrt != nil {
		ids := make([]int32, len(nrt.Resources))
		for i, r := range nrt.Resources {
			cs := C.CString(r)
			ids[i] = int32(C.spx_ingest_resource_id(in.h, cs))
			C.free(unsafe.Pointer(cs))
		}
		idp, freeID := cArray(ids)
		defer freeID()
		wp, freeW := cArray(nrt.Weights)
		defer freeW()
		params = C.spx_nrt_params{strategy: C.int32_t(nrt.Strategy), n_weights: C.int32_t(len(ids)), weight_res: (*C.int32_t)(idp), weight: (*C.int64_t)(wp)}
		// (params lives in C-visible memory for the duration of the call only: cgo pins the Go struct behind the pointer passed below)
		o.nrt, o.nrt_params = C.spx_ingest_nrt_objects(in.h), &params
	}
	if network {
		o.appgroups, o.nettopo = C.spx_ingest_appgroup_objects(in.h), C.spx_ingest_nettopo_objects(in.h)
	}
	if quota {
		o.quota = C.spx_ingest_quota_objects(in.h)
	}
	e.mu.Lock() // a writer: nothing else may drive the engine while its tables are replaced
	defer e.mu.Unlock()
	if rc := C.spx_load_profile(e.h, &o); rc != 0 {
		return e.err("spx_load_profile")
	}
	return nil
}

// UploadFeasibleMask tells the engine which (pod, node) cells passed the Filter plugins that run OUTSIDE it (upstream's in-tree
// filters): NormalizeScore-type plugins then normalise over those cells only, as RunScorePlugins does.  mask[p*nNodes+n] != 0 = feasible.
func (e *Engine) UploadFeasibleMask(mask []uint8, nPods, nNodes int64) error {
	p, free := cArray(mask)
	defer free()
	if rc := C.spx_upload_feasible_mask(e.h, (*C.uint8_t)(p), C.int64_t(nPods), C.int64_t(nNodes)); rc != 0 {
		return e.err("spx_upload_feasible_mask")
	}
	return nil
}
