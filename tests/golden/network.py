"""Known answers of the reference's network-aware tests, as data.

pkg/networkaware/networkoverhead/networkoverhead_test.go: fixtures GetNetworkTopologyCRBasic (:188-224),
GetAppGroupCRBasic (:309-347; dependencies carry no MaxNetworkCost -> 0), nodes (:580-597, :1070-1087),
TestNetworkOverheadScore (:572-818), TestNetworkOverheadFilter (:1055-1276; weightsName "UserDefined").
pkg/networkaware/topologicalsort/topologicalsort_test.go:187-246 (Less)."""

NODES = [  # (name, region, zone)
    ("n-1", "us-west-1", "Z1"), ("n-2", "us-west-1", "Z1"), ("n-3", "us-west-1", "Z2"), ("n-4", "us-west-1", "Z2"),
    ("n-5", "us-east-1", "Z3"), ("n-6", "us-east-1", "Z3"), ("n-7", "us-east-1", "Z4"), ("n-8", "us-east-1", "Z4"),
]
REGION_COSTS = {"us-west-1": [("us-east-1", 20)], "us-east-1": [("us-west-1", 20)]}
ZONE_COSTS = {"Z1": [("Z2", 5)], "Z2": [("Z1", 5)], "Z3": [("Z4", 10)], "Z4": [("Z3", 10)]}
APPGROUP_BASIC = {
    "workloads": [{"selector": "p1", "dependencies": [("p2", 0)]}, {"selector": "p2", "dependencies": [("p3", 0)]},
                  {"selector": "p3", "dependencies": []}],
    "topology_order": [("p1", 1), ("p2", 2), ("p3", 3)],
}

SCORE_PLACED = [("p1", "n-2"), ("p2", "n-5"), ("p3", "n-1")]   # :620-624
SCORE_CASES = [
    dict(name="p1 to allocate, 8 nodes to score", line=615, selector="p1", appgroup="basic",
         before=[20, 20, 20, 20, 0, 1, 10, 10], after=[0, 0, 0, 0, 100, 95, 50, 50]),
    dict(name="p2 to allocate, 8 nodes to score", line=651, selector="p2", appgroup="basic",
         before=[0, 1, 5, 5, 20, 20, 20, 20], after=[100, 95, 75, 75, 0, 0, 0, 0]),
    dict(name="p3 to allocate, no dependency", line=686, selector="p3", appgroup="basic",
         before=[0] * 8, after=[0] * 8),
]

FILTER_PLACED = [("p1", "n-2"), ("p2", "n-5"), ("p3", "n-8")]  # :1063-1067
FILTER_CASES = [  # (line, selector, appgroup, node index, satisfied, violated) ; pass iff not violated > satisfied
    dict(line=1102, selector="p1", appgroup="basic", node=0, want=(0, 1)),   # "Satisfied: 0 Violated: 1"
    dict(line=1114, selector="p1", appgroup="basic", node=5, want=None),
    dict(line=1126, selector="p2", appgroup="basic", node=4, want=(0, 1)),
    dict(line=1138, selector="p2", appgroup="basic", node=6, want=None),
    dict(line=1150, selector="p3", appgroup="basic", node=0, want=None),
    dict(line=1162, selector="p10", appgroup="", node=0, want=None),
    dict(line=1174, selector="p1", appgroup="basic", node=0, want=(0, 1)),
    dict(line=1186, selector="p1", appgroup="basic", node=5, want=None),
]

# GetAppGroupCROnlineBoutique (topologicalsort_test.go / networkoverhead_test.go:226-307): 11 workloads
ONLINEBOUTIQUE = {
    "workloads": [
        {"selector": "p1", "dependencies": [(s, 0) for s in ("p2", "p3", "p4", "p6", "p8", "p9", "p10")]},
        {"selector": "p2", "dependencies": [("p11", 0)]},
        {"selector": "p3", "dependencies": []}, {"selector": "p4", "dependencies": []}, {"selector": "p5", "dependencies": []},
        {"selector": "p6", "dependencies": []}, {"selector": "p7", "dependencies": []},
        {"selector": "p8", "dependencies": [(s, 0) for s in ("p2", "p3", "p4", "p5", "p6", "p7")]},
        {"selector": "p9", "dependencies": [("p3", 0)]},
        {"selector": "p10", "dependencies": []}, {"selector": "p11", "dependencies": []},
    ],
    # Status.TopologyOrder as written in the fixture; the test sorts it by selector before use (:261)
    "topology_order": [("p1", 1), ("p10", 2), ("p9", 3), ("p8", 4), ("p7", 5), ("p6", 6), ("p5", 7), ("p4", 8), ("p3", 9),
                       ("p2", 10), ("p11", 11)],
}

# TestTopologicalSortLess topologicalsort_test.go:187-246: (appgroup1, selector1, appgroup2, selector2, want)
LESS_CASES = [
    dict(line=187, p1=("basic", "p1"), p2=("basic", "p2"), want=True),
    dict(line=207, p1=("onlineboutique", "p5"), p2=("onlineboutique", "p1"), want=False),
    dict(line=228, p1=("basic", "p1"), p2=("other", "p5"), want=False),  # equal priority and timestamp -> PrioritySort false
]

# test/integration/topologicalsort_test.go:253-342: pods of ONE AppGroup created with equal priority are popped from
# the queue in Status.TopologyOrder index order (the fixture's indexes, after the by-selector sort at :248-249).
QUEUE_ORDER_CASES = [
    dict(line=255, appgroup="basic", created=["p1", "p2"], popped=["p1", "p2"]),
    dict(line=266, appgroup="basic", created=["p1", "p3"], popped=["p1", "p3"]),
    dict(line=277, appgroup="basic", created=["p2", "p3"], popped=["p2", "p3"]),
    dict(line=288, appgroup="basic", created=["p1", "p2", "p3"], popped=["p1", "p2", "p3"]),
    dict(line=301, appgroup="onlineboutique", created=["p1", "p5"], popped=["p1", "p5"]),
    dict(line=312, appgroup="onlineboutique", created=["p4", "p8"], popped=["p8", "p4"]),
    dict(line=323, appgroup="onlineboutique", created=[f"p{i}" for i in range(1, 12)],
         popped=["p1", "p10", "p9", "p8", "p7", "p6", "p5", "p4", "p3", "p2", "p11"]),
]
