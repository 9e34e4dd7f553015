"""Placements expected by the reference's integration tests (test/integration/*.go) for the plugins on the hot path, as
data.  These run a whole scheduler upstream; what they pin for this engine is the score ordering per pod and, for the
trimaran ones, the one-pod-at-a-time sequence (the second pod sees the first one bound).

test/integration/targetloadpacking_test.go:55-200           TargetLoadPacking only, default args (target 40, multiplier 1.5)
test/integration/loadVariationRiskBalancing_test.go:55-195  LoadVariationRiskBalancing only, default args
test/integration/allocatable_test.go:44-135                 NodeResourcesAllocatable, memory weight 10, Least / Most; the
                                                            expected node per pod is a SET (selectHost draws among ties)
"""

TLP = dict(
    line=55, window_end=0,
    nodes=[dict(name="node-1", allocatable={"pods": "32", "cpu": "2", "memory": "256"}, capacity={"pods": "32", "cpu": "2", "memory": "256"}),
           dict(name="node-2", allocatable={"pods": "32", "cpu": "2", "memory": "256"}, capacity={"pods": "32", "cpu": "2", "memory": "256"}),
           dict(name="node-3", allocatable={"pods": "32", "cpu": "2", "memory": "256"}, capacity={"pods": "32", "cpu": "2", "memory": "256"})],
    metrics={0: [("CPU", "Latest", 10.0)], 1: [("CPU", "Latest", 60.0)], 2: [("CPU", "Latest", 0.0)]},
    pods=[{"cpu": "300m", "memory": "50"}, {"cpu": "100m", "memory": "50"}],  # requests only
    expected=["node-1", "node-1"],
)

LVRB = dict(
    line=55, window_end=1556985422,
    nodes=[dict(name=f"node-{i}", allocatable={"cpu": "2", "memory": "256"}, capacity={"cpu": "2", "memory": "256"}) for i in (1, 2, 3)],
    metrics={0: [("CPU", "AVG", 30.0)], 1: [("CPU", "AVG", 70.0), ("CPU", "STD", 20.0)], 2: [("CPU", "AVG", 40.0), ("CPU", "STD", 30.0)]},
    pods=[{"cpu": "300m", "memory": "50"}, {"cpu": "100m", "memory": "50"}],
    expected=["node-1", "node-1"],
)

_SMALL = {"pods": "32", "cpu": "500m", "memory": "500"}
_BIG = {"pods": "32", "cpu": "500m", "memory": "5000"}
ALLOCATABLE = dict(
    line=44, weights={"memory": 10},
    nodes=[dict(name="fake-node-small-1", allocatable=_SMALL, capacity=_SMALL), dict(name="fake-node-small-2", allocatable=_SMALL, capacity=_SMALL),
           dict(name="fake-node-big", allocatable=_BIG, capacity=_BIG)],
    pods=[("small-1", {"memory": "100"}), ("small-2", {"memory": "100"}), ("small-3", {"memory": "100"}), ("small-4", {"memory": "100"}),
          ("big-1", {"memory": "2000"})],
    # big-1 does not fit the small nodes (upstream's NodeResourcesFit, outside this engine): the harness masks them out
    expected={"Least": {"small-1": {"fake-node-small-1", "fake-node-small-2"}, "small-2": {"fake-node-small-1", "fake-node-small-2"},
                        "small-3": {"fake-node-small-1", "fake-node-small-2"}, "small-4": {"fake-node-small-1", "fake-node-small-2"},
                        "big-1": {"fake-node-big"}},
              "Most": {p: {"fake-node-big"} for p in ("small-1", "small-2", "small-3", "small-4", "big-1")}},
)

# test/integration/lowriskovercommitment_test.go:48-209: LowRiskOverCommitment only, SmoothingWindowSize default (5),
# RiskLimitWeights cpu = memory = 1 (the limit risk alone decides).  pod-1 and pod-2 already run on node-1 / node-2; pod-3 must go to
# node-1: there the limits stay under the capacity (risk 0, score 100), on node-2 they exceed it by 200m out of a 1600m gap (88).
LROC = dict(
    line=48, params=dict(smoothing_window_size=5, w_cpu=1.0, w_mem=1.0),
    nodes=[dict(name="node-1", allocatable={"cpu": "2", "memory": "256"}, capacity={"cpu": "2", "memory": "256"}),
           dict(name="node-2", allocatable={"cpu": "2", "memory": "256"}, capacity={"cpu": "2", "memory": "256"})],
    metrics={0: [("CPU", "AVG", 60.0), ("CPU", "STD", 30.0)], 1: [("CPU", "AVG", 30.0), ("CPU", "STD", 20.0)]},
    on_node={0: [({"cpu": "500m", "memory": "64"}, {"cpu": "500m", "memory": "64"})],
             1: [({"cpu": "100m", "memory": "64"}, {"cpu": "1200m", "memory": "64"})]},
    pod=({"cpu": "500m", "memory": "64"}, {"cpu": "1000m", "memory": "64"}),
    expected="node-1", scores=[100, 88],
)

# test/integration/peaks_test.go:44-209: Peaks only; three idle nodes (2 cpus) whose power models differ in K1
# (-91.5, -1091.5, -2091.5; same K0, K2).  pod-1 (300m) takes the node with the smallest power jump, node-1.  pod-2 (1900m) no
# longer fits node-1 next to pod-1 — that is upstream's NodeResourcesFit, outside this engine, so the harness masks node-1 out —
# and takes node-2.
PEAKS = dict(
    line=44,
    nodes=[dict(name=f"node-{i}", allocatable={"pods": "32", "cpu": "2", "memory": "256"}, capacity={"pods": "32", "cpu": "2", "memory": "256"}) for i in (1, 2, 3)],
    models=[{"k0": 471.7412504314313, "k1": k1, "k2": -0.07186049052516228} for k1 in (-91.50493019588365, -1091.50493019588365, -2091.50493019588365)],
    metrics={0: [("CPU", "Latest", 0.0)], 1: [("CPU", "Latest", 0.0)], 2: [("CPU", "Latest", 0.0)]},
    pods=[{"cpu": "300m", "memory": "50"}, {"cpu": "1900m", "memory": "50"}],
    feasible=[[1, 1, 1], [0, 1, 1]],
    expected=["node-1", "node-2"],
)
