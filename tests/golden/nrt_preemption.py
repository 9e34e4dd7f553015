"""Known answers of the reference's NRT eviction-simulation test, as data.

pkg/noderesourcetopology/preemption/preemption_test.go: TestGetNRTPostPodsEviction (:32-378, 7 cases), fixtures getTestNRT
(:382-432) and getTestEncodedInfo10Containers (:436-484: container -> NUMA node affinities)."""

def _zone(name, extra, avail_cpu="1", avail_mem="100Mi", avail_dev="1"):
    return {"name": name, "type": "Node", "resources": [("cpu", "10", "10", avail_cpu), ("memory", "500Mi", "500Mi", avail_mem),
                                                        (extra, "8", "8", avail_dev)]}


DEV_A, DEV_B = "example-device.com/deviceA", "example-device.com/deviceB"
TEST_NRT = {"zones": [_zone("node-0", DEV_A, avail_dev="1"), _zone("node-1", DEV_B, avail_dev="2")]}   # getTestNRT
# (namespace, pod, container) -> NUMA node
PLACEMENT = {("ns-a", "pod-0", "cnt-0"): 0, ("ns-a", "pod-0", "cnt-1"): 0, ("ns-a", "pod-0", "cnt-3"): 0, ("ns-a", "pod-a-1", "cnt-0"): 0,
             ("ns-a", "pod-a-2", "cnt-0"): 0, ("ns-b", "pod-0", "cnt-0"): 1, ("ns-b", "pod-0", "cnt-1"): 1, ("ns-b", "pod-0", "cnt-3"): 1,
             ("ns-b", "pod-1", "cnt-0"): 1, ("ns-b", "pod-2", "cnt-0"): 1}

G, BU, BE = 0, 1, 2  # SPX_QOS_GUARANTEED / BURSTABLE / BESTEFFORT (Status.QOSClass of the fixture pods)


def ctr(name, requests=None, limits=None):
    return {"name": name, "requests": requests or {}, "limits": limits or {}}


_ONE = {"cpu": "1", "memory": "100Mi", DEV_A: "1"}
CASES = [
    dict(name="no victims", line=42, victims=[], placement=PLACEMENT, error="no victims found, cannot process eviction simulation"),
    dict(name="empty numa placement info with victims", line=50, placement=None,
         victims=[dict(ns="", name="pod-0", qos=G, containers=[ctr("container-0", _ONE)])],
         error="numa placement info not found, cannot process eviction simulation"),
    dict(name="victims with non-exclusive resources", line=80, placement=PLACEMENT,
         victims=[dict(ns="", name="pod-0", qos=BE, containers=[ctr("container-0")])],
         error="no resources to add, cannot process eviction simulation"),
    dict(name="mixed victims with exclusive resources", line=104, placement=PLACEMENT, error="",
         victims=[
             dict(ns="ns-a", name="pod-0", qos=BU, containers=[
                 ctr("cnt-0", {"cpu": "1", "memory": "100Mi", DEV_A: "1"}, {"cpu": "2", "memory": "100Mi", DEV_A: "1"}),
                 ctr("cnt-1", {"cpu": "1", "memory": "100Mi", DEV_A: "2"}, {"cpu": "2", "memory": "200Mi", DEV_A: "2"}),
                 ctr("cnt-2", {"cpu": "1", "memory": "100Mi"}, {"cpu": "2", "memory": "200Mi"}),
                 ctr("cnt-3", {"cpu": "1", "memory": "100Mi", DEV_A: "1"}, {"cpu": "2", "memory": "200Mi", DEV_A: "1"})]),
             dict(ns="ns-b", name="pod-1", qos=BE, containers=[ctr("cnt-0", {DEV_B: "3"})]),
             dict(ns="ns-b", name="pod-2", qos=G, containers=[ctr("cnt-0", {DEV_B: "3", "cpu": "2", "memory": "100Mi"})]),
         ],
         # :224-276: node-0 deviceA 1 -> 5; node-1 cpu 1 -> 3, memory 100Mi -> 200Mi ("only for the guaranteed pod containers"), deviceB 2 -> 8
         expected={"zones": [_zone("node-0", DEV_A, avail_dev="5"), _zone("node-1", DEV_B, avail_cpu="3", avail_mem="200Mi", avail_dev="8")]}),
    dict(name="victim not found in numa placement info", line=279, placement=PLACEMENT,
         victims=[dict(ns="ns-a", name="newpod", qos=G, containers=[ctr("cnt-0", _ONE)])],
         error="no resources to add, cannot process eviction simulation"),
    dict(name="resources release exceeds allocatable", line=311, placement=PLACEMENT,
         victims=[dict(ns="ns-a", name="pod-0", qos=G, containers=[ctr("cnt-0", {"cpu": "15", "memory": "100Mi", DEV_A: "20"})])],
         error="resource release request exceeds NUMA allocatable"),
    dict(name="numa placement info with no containers", line=343, placement={},
         victims=[dict(ns="", name="pod-0", qos=BE, containers=[])],
         error="no containers found in numa placement info, cannot process eviction simulation"),
]
ERROR_CODES = {  # include/spx.h SPX_EVICT_*
    "": 0, "NRT not found, cannot process eviction simulation": 1, "no victims found, cannot process eviction simulation": 2,
    "numa placement info not found, cannot process eviction simulation": 3,
    "no containers found in numa placement info, cannot process eviction simulation": 4,
    "no resources to add, cannot process eviction simulation": 5, "resource release request exceeds NUMA allocatable": 6,
}
