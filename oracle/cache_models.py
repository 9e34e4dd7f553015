"""CPU restatements of two control-plane caches whose VERDICTS feed the hot path — TEST INFRASTRUCTURE (as oracle/pyoracle.py:
only tests/ may import this).  Pure Python, small cases only.  Pinned by the reference's own tests, transcribed as data:
tests/golden/nrt_discard_reserved.json (discardreserved_test.go:34-140), tests/golden/trimaran_handler.json (handler_test.go:12-77).

The caches themselves stay in Go (control plane); what crosses the boundary is
  * DiscardReserved -> CachedNRTInfo.Fresh per node (the `fresh` column of spx_nrt_objects) and "no NRT",
  * PodAssignEventHandler -> the per-node list of (pod, bind time) pairs (spx_assigned_objects) TargetLoadPacking's Score walks.
"""
from __future__ import annotations

import bisect
from typing import Dict, List, Optional, Tuple


class DiscardReservedModel:
    """pkg/noderesourcetopology/cache/discardreserved.go:47-115"""

    def __init__(self, api_nrts: Optional[Dict[str, object]] = None):
        self.reservation_map: Dict[str, Dict[str, bool]] = {}  # node -> pod UID -> true            (:49)
        self.api = dict(api_nrts or {})                         # what client.Get would return       (:73)

    def get_cached_nrt_copy(self, node: str) -> Tuple[Optional[object], bool]:
        """(nrt or None, CachedNRTInfo.Fresh)  :62-76"""
        if len(self.reservation_map.get(node, {})) > 0:  # :65-69: any reservation -> (nil, CachedNRTInfo{}) i.e. Fresh == false
            return None, False
        return self.api.get(node), True                  # :71-76: a failed Get still answers Fresh == true with a nil object

    def reserve(self, node: str, uid: str) -> None:      # ReserveNodeResources :86-95
        self.reservation_map.setdefault(node, {})[uid] = True

    def remove_reservation(self, node: str, uid: str) -> None:  # Unreserve / PostBind -> removeReservationForNode :97-115
        self.reservation_map.get(node, {}).pop(uid, None)       # (delete on a nil inner map is a no-op in Go)


class PodAssignHandlerModel:
    """pkg/trimaran/handler.go:60-170: ScheduledPodsCache[node] = [(timestamp, pod)], appended in arrival order"""

    def __init__(self, reporting_interval_s: int = 60):      # metricsAgentReportingIntervalSeconds
        self.cache: Dict[str, List[Tuple[Optional[float], str]]] = {}
        self.interval = reporting_interval_s

    def on_update(self, old_node: str, new_node: str, pod: str, now: float) -> None:
        if old_node != new_node:                              # OnUpdate :108-115
            self.update_cache(new_node, pod, now)

    def update_cache(self, node: str, pod: str, now: float) -> None:
        if node == "":                                        # updateCache :138-146: unassigned pods are not cached
            return
        self.cache.setdefault(node, []).append((now, pod))

    def cleanup(self, now: float) -> None:
        """cleanupCache :149-170.  sort.Search is a BINARY search for the first entry younger than the interval — it assumes the
        list is ordered by time, which holds for appended entries; an entry with the zero time.Time (None here) is older than
        anything.  When no entry is young enough (idx == len) the node's list is left as it is (:158-160)."""
        for node in list(self.cache):
            lst = self.cache[node]
            young = [ts is not None and ts + self.interval > now for ts, _ in lst]
            idx = bisect.bisect_left(young, True)             # sort.Search over the predicate, as Go evaluates it
            if idx == len(lst):
                continue
            self.cache[node] = lst[idx:]
            if not self.cache[node]:
                del self.cache[node]

    def pods(self, node: str) -> List[str]:
        return [p for _, p in self.cache.get(node, [])]
