"""Known answers of the reference's helper-level tests on the NodeResourceTopologyMatch / effective-request path,
as data (hand-transcribed; inputs and expected outputs only).

pkg/util/resource_test.go:34-149              TestGetPodEffectiveRequest (cpu in millicores, memory in bytes)
pkg/noderesourcetopology/numaresources_test.go:29-115   TestIsHostLevelResource / TestIsNUMAAffineResource
pkg/noderesourcetopology/numaresources_test.go:117-373  TestSubtractResourcesFromNUMANodeList
pkg/noderesourcetopology/numaresources_test.go:375-462  TestSubstractNUMA
pkg/noderesourcetopology/pluginhelpers_test.go:29-107   TestOnlyNonNUMAResources
pkg/noderesourcetopology/nodeconfig/topologymanager_test.go:256-428, :430-498, :500-607
                                              TestConfigFromAttributes / TestConfigFromPolicies / TestConfigFromNRT
pkg/noderesourcetopology/least_numa_test.go:758-920           TestMinDistance (minAvgDistanceInCombinations, float32)
pkg/noderesourcetopology/cache/store_test.go:998-1108         TestResourceStoreUpdate
pkg/noderesourcetopology/cache/overreserve_test.go:292-342    TestGetCachedNRTCopyReserve (topology: cache_test.go:282-309)
"""

# (line, app container requests [(cpu milli, mem bytes)], init container requests, overhead or None, want (cpu, mem))
EFFECTIVE_REQUEST = [
    (43, [(1, 1)], [], None, (1, 1)),
    (51, [(1, 1), (2, 3)], [], None, (3, 4)),
    (60, [(1, 1), (2, 3)], [(1, 1)], None, (3, 4)),
    (71, [(1, 1), (2, 3)], [(10, 1)], None, (10, 4)),
    (82, [(1, 1), (2, 3)], [(10, 1), (1, 10)], None, (10, 10)),
    (94, [(1, 1), (2, 3)], [(10, 1), (1, 1)], None, (10, 4)),
    (106, [(1, 1)], [], (1, 1), (2, 2)),
    (115, [(1, 1), (2, 3)], [(1, 1)], (1, 1), (4, 5)),
]

# resource name -> (isHostLevelResource, isNUMAAffineResource)   numaresources_test.go:34-61, :78-105
RESOURCE_CLASSES = {
    "cpu": (False, True),
    "memory": (False, True),
    "hugepages-1Gi": (False, True),
    "storage": (True, False),
    "ephemeral-storage": (True, False),
    "vendor.io/fastest-nic": (True, False),
    "awesome.com/gpu-for-ai": (True, False),
}

# TestSubtractResourcesFromNUMANodeList: zones = [(numa id, {resource: quantity})]; expected None = the reference returns
# an error (and the list is not compared)
SUBTRACT_NUMA = [
    dict(line=128, name="empty from empty", zones=[(0, {})], numa_id=0, qos="Guaranteed", request={}, expected=[(0, {})]),
    dict(line=146, name="inconsistent numaID", zones=[(0, {})], numa_id=2, qos="Guaranteed", request={}, expected=[(0, {})]),
    dict(line=164, name="empty from minimal", zones=[(0, {"cpu": "2", "memory": "4Gi"})], numa_id=0, qos="Guaranteed", request={},
         expected=[(0, {"cpu": "2", "memory": "4Gi"})]),
    dict(line=188, name="remove core resources (GU qos)", zones=[(0, {"cpu": "8", "memory": "16Gi"})], numa_id=0, qos="Guaranteed",
         request={"cpu": "2", "memory": "4Gi"}, expected=[(0, {"cpu": "6", "memory": "12Gi"})]),
    dict(line=215, name="remove only devices resources (BU qos)", zones=[(0, {"cpu": "8", "memory": "16Gi", "vendor.io/gpu": "4"})],
         numa_id=0, qos="Burstable", request={"cpu": "2", "memory": "4Gi", "vendor.io/gpu": "2"},
         expected=[(0, {"cpu": "8", "memory": "16Gi", "vendor.io/gpu": "2"})]),
    dict(line=245, name="skip hostlevel resources (GU qos)", zones=[(0, {"cpu": "8", "memory": "16Gi", "vendor.io/nic": "4"})],
         numa_id=0, qos="Guaranteed", request={"cpu": "6", "memory": "12Gi", "vendor.io/nic": "2", "ephemeral-storage": "1Gi"},
         expected=[(0, {"cpu": "2", "memory": "4Gi", "vendor.io/nic": "2"})]),
    dict(line=276, name="remove excessive core resources (GU qos)", zones=[(0, {"cpu": "8", "memory": "16Gi"})], numa_id=0,
         qos="Guaranteed", request={"cpu": "10", "memory": "20Gi"}, expected=None),
    dict(line=304, name="require missing resources (GU qos, device)", zones=[(0, {"cpu": "8", "memory": "16Gi"})], numa_id=0,
         qos="Guaranteed", request={"cpu": "4", "memory": "8Gi", "vendor.io/gpu": "2"}, expected=[(0, {"cpu": "4", "memory": "8Gi"})]),
    dict(line=332, name="require missing resources (GU qos, core)", zones=[(0, {"cpu": "8", "memory": "16Gi"})], numa_id=0,
         qos="Guaranteed", request={"cpu": "4", "memory": "8Gi", "hugepages-1Gi": "2Gi"}, expected=[(0, {"cpu": "4", "memory": "8Gi"})]),
]

# TestSubstractNUMA: subtractFromNUMAs(resources, numaNodes, nodes...)
SUBTRACT_NUMAS = [
    dict(line=384, name="simple", zones=[(0, {"cpu": "8", "memory": "10Gi"})], request={"cpu": "2", "memory": "2Gi"}, nodes=[0],
         expected=[(0, {"cpu": "6", "memory": "8Gi"})]),
    dict(line=410, name="substract resources from 2 NUMA nodes",
         zones=[(0, {"cpu": "8", "memory": "10Gi"}), (1, {"cpu": "8", "memory": "10Gi"})], request={"cpu": "12", "memory": "2Gi"},
         nodes=[0, 1], expected=[(0, {"cpu": "0", "memory": "8Gi"}), (1, {"cpu": "4", "memory": "10Gi"})]),
]

# TestOnlyNonNUMAResources pluginhelpers_test.go:30-47 (zones), :53-96 (cases)
ONLY_NON_NUMA_ZONES = [(0, {"cpu": "8", "memory": "10Gi", "gpu": "1"}), (1, {"cpu": "8", "memory": "10Gi", "nic": "1"})]
ONLY_NON_NUMA = [
    (54, {"resource1": "1", "resource2": "1"}, True),
    (62, {"cpu": "1"}, False),
    (69, {"cpu": "1", "memory": "1"}, False),
    (77, {"gpu": "1"}, False),
    (84, {"nic": "1"}, False),
    (91, {"nic": "1", "gpu": "1"}, False),
]

# TestConfigFromNRT (:500-607): TopologyPolicies, Attributes -> (policy, scope, MaxNUMANodes); defaults = (none, container, 8)
CONFIG_FROM_NRT = [
    (507, [], {}, ("none", "container", 8)),
    (512, ["BestEffortPodLevel"], {}, ("best-effort", "pod", 8)),
    (525, ["RestrictedContainerLevel", "BestEffortPodLevel"], {}, ("restricted", "container", 8)),
    (539, [], {"topologyManagerPolicy": "restricted"}, ("restricted", "container", 8)),
    (555, ["BestEffortPodLevel"], {"topologyManagerScope": "container"}, ("best-effort", "container", 8)),
    (574, ["BestEffortPodLevel"], {"topologyManagerScope": "container", "topologyManagerPolicy": "restricted"},
     ("restricted", "container", 8)),
]
# TestConfigFromAttributes (:256-428), applied on top of the defaults: invalid values are ignored, MaxNUMANodes is capped
CONFIG_FROM_ATTRIBUTES = [
    (268, {}, ("none", "container", 8)),  # (:263 is the same with a nil list)
    (273, {"topologyManagerScope": "pod"}, ("none", "pod", 8)),
    (285, {"topologyManagerPolicy": "restricted"}, ("restricted", "container", 8)),
    (297, {"topologyManagerPolicy": "restricted", "topologyManagerScope": "container"}, ("restricted", "container", 8)),
    (314, {"topologyManagerScope": "pod", "topologyManagerPolicy": "single-numa-node"}, ("single-numa-node", "pod", 8)),
    (331, {"topologyManagerScope": "Pod", "topologyManagerPolicy": "single-numa-node"}, ("single-numa-node", "container", 8)),
    (347, {"topologyManagerScope": "Container", "topologyManagerPolicy": "restricted"}, ("restricted", "container", 8)),
    (363, {"topologyManagerMaxNUMANodes": "A"}, ("none", "container", 8)),
    (373, {"topologyManagerMaxNUMANodes": "0"}, ("none", "container", 8)),
    (383, {"topologyManagerMaxNUMANodes": "-2"}, ("none", "container", 8)),
    (393, {"topologyManagerMaxNUMANodes": "16"}, ("none", "container", 16)),
    (405, {"topologyManagerMaxNUMANodes": "65535"}, ("none", "container", 1024)),
]

# TestConfigFromPolicies (:430-498), on top of the defaults: only the first entry counts, unknown names are ignored
CONFIG_FROM_POLICIES = [
    (442, [], ("none", "container", 8)),  # (:437 is the same with a nil list)
    (447, ["SingleNUMANodePodLevel"], ("single-numa-node", "pod", 8)),
    (455, ["SingleNUMANodeContainerLevel"], ("single-numa-node", "container", 8)),
    (463, ["RestrictedContainerLevel"], ("restricted", "container", 8)),
    (471, ["RestrictedContainerLevel", "SingleNUMANodePodLevel"], ("restricted", "container", 8)),
    (482, ["foobar"], ("none", "container", 8)),
]

# OverReserve: zone Available minus the effective request of every pod assumed on the node, on EVERY zone that reports
# the resource (resourceStore.UpdateNRT).  zones = [(id, {resource: available})], pods = containers' requests
OVER_RESERVE = [
    dict(line=998, source="cache/store_test.go", zones=[(0, {"cpu": "20", "memory": "32Gi"}), (1, {"cpu": "20", "memory": "32Gi", "vendor.com/nic": "8"})],
         assumed_pods=[[{"cpu": "16", "memory": "4Gi", "vendor.com/nic": "2"}, {"cpu": "2", "memory": "2Gi"}]],
         expected=[(0, {"cpu": "2", "memory": "26Gi"}), (1, {"cpu": "2", "memory": "26Gi", "vendor.com/nic": "6"})]),
    dict(line=292, source="cache/overreserve_test.go",
         zones=[(0, {"cpu": "30", "memory": "60Gi", "vendor.com/nic": "16"}), (1, {"cpu": "30", "memory": "60Gi", "vendor.com/nic": "16"})],
         assumed_pods=[[{"cpu": "8", "memory": "16Gi"}]],
         expected=[(0, {"cpu": "22", "memory": "44Gi", "vendor.com/nic": "16"}), (1, {"cpu": "22", "memory": "44Gi", "vendor.com/nic": "16"})]),
]

# TestMinDistance (least_numa_test.go:758-920): zone cost maps and the minimal average distance per subset size
# (float32; 255 per missing entry).  The "three numa node" case lists 3 of the 4 subsets; the fourth has the same average.
MIN_DISTANCE_COSTS = {0: {0: 10, 1: 12, 2: 20, 3: 20}, 1: {0: 12, 1: 10, 2: 20, 3: 20}, 2: {0: 20, 1: 20, 2: 10, 3: 12},
                      3: {0: 20, 1: 20, 2: 12, 3: 10}}
MIN_DISTANCE = [  # (line, with costs?, subset size, expected)
    (820, True, 1, 10.0),
    (838, True, 2, 11.0),
    (861, True, 3, 14.888889),
    (878, False, 2, 255.0),
]


# TestIncludeNonNative  pkg/noderesourcetopology/resourcerequests/exclusive_test.go:36-46 over coreTestCases (:174-437):
# (line, name, app containers, init containers, sidecar init containers, expectedNonNative); a container = its requests
# (limits equal the requests in every case).  expectedExclusive (AreExclusiveForPod) belongs to the cache's foreign-pod
# tracking, which is outside the scored path.
_GU = {"cpu": "4", "memory": "2Gi"}
_FPGA = {"veryfast.io/fpga": "1"}
INCLUDE_NON_NATIVE = [
    (178, "no containers", [], [], [], False),
    (189, "single-container-gu-no-devs", [_GU], [], [], False),
    (217, "single-initcontainer-gu-no-devs", [], [_GU], [], False),
    (245, "single-sidecar-initcontainer-gu-no-devs", [], [], [_GU], False),
    (274, "single-container-devs-only", [_FPGA], [], [], True),
    (300, "single-initcontainer-devs-only", [], [_FPGA], [], True),
    (326, "single-sidecar-initcontainer-devs-only", [], [], [_FPGA], True),
    (353, "single-container-gu-core-and-devs", [{"cpu": "8", "memory": "16Gi", "veryfast.io/fpga": "1"}], [], [], True),
    (383, "single-container-nongu-cpus-and-devs", [{"cpu": "8", "veryfast.io/fpga": "1"}], [], [], True),
    (411, "single-container-nongu-cpus-only", [{"cpu": "8"}], [], [], False),
]
