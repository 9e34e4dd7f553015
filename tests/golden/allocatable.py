"""pkg/noderesources/allocatable_test.go:114-238 (TestNodeResourcesAllocatable), as data.

node = (milliCPU, memory) as makeNodeInfo(node, milliCPU, memory) builds it (:315-331);
`expected` is the NodeScoreList after NormalizeScore (:289-306)."""

DEFAULT = {"cpu": 1 << 20, "memory": 1}   # defaultResourceAllocatableSet :93-96
CPU_HEAVY = {"cpu": 1 << 30, "memory": 1}  # cpuResourceAllocatableSet :99-102
MIB = 1 << 20

NO_RESOURCES = {"containers": []}
CPU_AND_MEMORY = {"containers": [{"requests": {"cpu": "1000m", "memory": "1Gi"}}]}  # :82-85
BIG_CPU = {"containers": [{"requests": {"cpu": "8000m", "memory": "1Gi"}}]}         # :86-89

CASES = [
    dict(name="nothing scheduled, nothing requested", line=115, pod=NO_RESOURCES,
         nodes=[(4000, 10000), (4000, 10000)], resources=DEFAULT, mode="Least", expected=[0, 0]),
    dict(name="differently sized machines, least mode", line=122, pod=CPU_AND_MEMORY,
         nodes=[(4000, 10000), (6000, 10000)], resources=DEFAULT, mode="Least", expected=[100, 0]),
    dict(name="differently sized machines, most mode", line=129, pod=CPU_AND_MEMORY,
         nodes=[(4000, 10000), (6000, 10000)], resources=DEFAULT, mode="Most", expected=[0, 100]),
    dict(name="no resources requested, pods scheduled", line=136, pod=NO_RESOURCES,
         nodes=[(4000, 10000), (4000, 10000)], resources=DEFAULT, mode="Least", expected=[0, 0]),
    dict(name="no resources requested, pods scheduled with resources", line=149, pod=NO_RESOURCES,
         nodes=[(10000, 20000), (10000, 20000)], resources=DEFAULT, mode="Least", expected=[0, 0]),
    dict(name="resources requested, pods scheduled with resources", line=159, pod=CPU_AND_MEMORY,
         nodes=[(10000, 20000), (10000, 20000)], resources=DEFAULT, mode="Least", expected=[0, 0]),
    dict(name="more than the node, least mode", line=169, pod=BIG_CPU,
         nodes=[(4000, 1000), (5000, 1000)], resources=DEFAULT, mode="Least", expected=[100, 0]),
    dict(name="more than the node, most mode", line=176, pod=BIG_CPU,
         nodes=[(4000, 1000), (5000, 1000)], resources=DEFAULT, mode="Most", expected=[0, 100]),
    dict(name="cpu weighted (Least)", line=183, pod=CPU_AND_MEMORY,
         nodes=[(1000, 2000), (1005, 1000)], resources=CPU_HEAVY, mode="Least", expected=[100, 0]),
    dict(name="cpu weighted (Most)", line=190, pod=CPU_AND_MEMORY,
         nodes=[(1000, 2000), (1005, 1000)], resources=CPU_HEAVY, mode="Most", expected=[0, 100]),
    dict(name="3 differently sized machines, least mode", line=197, pod=CPU_AND_MEMORY,
         nodes=[(1000, 1000 * MIB), (2000, 2000 * MIB), (3000, 3000 * MIB)], resources=DEFAULT, mode="Least",
         expected=[100, 50, 0]),
    dict(name="3 differently sized machines, most mode", line=210, pod=CPU_AND_MEMORY,
         nodes=[(1000, 1000 * MIB), (2000, 2000 * MIB), (3000, 3000 * MIB)], resources=DEFAULT, mode="Most",
         expected=[0, 50, 100]),
]

# validation errors :223-237 — weight <= 0 is rejected at construction (allocatable.go:53-61)
INVALID = [
    dict(name="resource with negative weight", line=224, resources={"memory": -1, "cpu": 1}),
    dict(name="resource with zero weight", line=232, resources={"memory": 1, "cpu": 0}),
]
