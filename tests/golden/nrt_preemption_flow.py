"""Known answers of the reference's Filter-inside-a-preemption-dry-run test, as data.

pkg/noderesourcetopology/filter_preemption_test.go: TestFilter_PreemptionFlow (:184-290, 7 sub-tests) on makePreemptionNRT
(:90-117: two NUMA zones, cpu 4 / memory 8Gi each; available cpu 0 and 2, memory 7Gi and 8Gi; SingleNUMANodeContainerLevel),
makeGuaranteedPod (:60-79) and makeEncodedInfoForPod (:130-166: every container of the pod on the given NUMA node).
The dispatch under test is filter.go:205-220: victims on the cycle state's stack (PreFilter/RemovePod, prefilter.go:51-101) and a
non-empty NUMA placement record -> the Filter runs on preemption.GetNRTPostPodsEviction's zone table, a failed simulation is
Unschedulable with the simulation's message; otherwise the ordinary Filter."""

NRT = {"zones": [{"name": "node-0", "type": "Node", "resources": [("cpu", "4", "4", "0"), ("memory", "8Gi", "8Gi", "7Gi")]},
                 {"name": "node-1", "type": "Node", "resources": [("cpu", "4", "4", "2"), ("memory", "8Gi", "8Gi", "8Gi")]}],
       "policies": ["SingleNUMANodeContainerLevel"]}
NODE = {"cpu": "8", "memory": "16Gi"}   # makeNodeFromNRT: the zones' capacities summed


def guaranteed(ns, name, cpu, memory):
    r = {"cpu": str(cpu), "memory": memory}
    return dict(ns=ns, name=name, containers=[dict(name="cnt", requests=r, limits=r)])


VICTIM = guaranteed("default", "victim", 4, "1Gi")
PLACEMENT = {("default", "victim", "cnt"): 0}      # makeEncodedInfoForPod(victim, 0)
ALIGN = "cannot align container"
CASES = [
    dict(line=197, name="without preemption (default mode) pod remains unschedulable", enabled=False, victims=[], placement=PLACEMENT,
         preemptor=guaranteed("default", "preemptor", 4, "1Gi"), want=ALIGN, over_reserved=True),
    dict(line=210, name="without preemption victims pod is unschedulable", enabled=True, victims=[], placement=PLACEMENT,
         preemptor=guaranteed("default", "preemptor", 4, "1Gi"), want=ALIGN, over_reserved=True),
    dict(line=223, name="victim eviction makes node schedulable", enabled=True, victims=[VICTIM], placement=PLACEMENT,
         preemptor=guaranteed("default", "preemptor", 4, "1Gi"), want=None, over_reserved=False),
    dict(line=237, name="victims without NUMA placement info preserve the normal Filter behavior", enabled=True, victims=[VICTIM],
         placement=None, preemptor=guaranteed("default", "preemptor", 4, "1Gi"), want=ALIGN, over_reserved=None),
    dict(line=248, name="preemption mode is disabled: no preemption effect", enabled=False, victims=[VICTIM], placement=None,
         preemptor=guaranteed("default", "preemptor", 4, "1Gi"), want=ALIGN, over_reserved=None),
    dict(line=259, name="eviction simulation failure results in unschedulable", enabled=True,
         victims=[guaranteed("ns-2", "victim2", 4, "1Gi")], placement=PLACEMENT, preemptor=guaranteed("default", "preemptor", 4, "1Gi"),
         want="eviction simulation in NRT is not possible:no resources to add, cannot process eviction simulation", over_reserved=False),
    dict(line=276, name="preemption with unschedulable result skips over-reserve marking", enabled=True, victims=[VICTIM],
         placement=PLACEMENT, preemptor=guaranteed("default", "too-large", 8, "1Gi"), want=ALIGN, over_reserved=False),
]
