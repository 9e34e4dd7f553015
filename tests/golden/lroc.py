"""Known answers of the reference's LowRiskOverCommitment tests, as data.

pkg/trimaran/lowriskovercommitment/beta_test.go: TestBetaDistribution_MatchMoments (:111-171),
TestBetaDistribution_DistributionFunction (:236-327, tolerance 1e-6 at :24), TestGetMaxVariance (:329-374).
pkg/trimaran/lowriskovercommitment/lowriskovercommitment_test.go: TestLowRiskOverCommitment_Score (:138-243),
fixtures node_A / watcherData_A / nrla_A1 / nrla_A2 (:261-338), TestLowRiskOverCommitment_computeRisk (:341-395).
pkg/trimaran/resourcestats_test.go: TestGetResourceLimits (:203-257), TestGetNodeRequestsAndLimits (:374-603),
pod builders getPodWithContainersAndOverhead / getPodWithLimits (:605-648)."""

TOLERANCE = 1e-6

MATCH_MOMENTS = [  # (name, m1, m2, want, alpha, beta)
    ("beta(1,1)", 0.5, 1.0 / 3.0, True, 1.0, 1.0),
    ("beta(0,0)", 0.0, 0.0, False, None, None),
]
DISTRIBUTION_FUNCTION = [  # (name, alpha, beta, x, want)
    ("beta(2,2) PDF(0.5)", 2, 2, 0.5, 0.5),
    ("beta(2,2) PDF(0.0)", 2, 2, 0.0, 0.0),
    ("beta(2,2) PDF(1.0)", 2, 2, 1.0, 1.0),
    ("beta(-1,1) PDF(0.5)", -1, 1, 0.5, 0.0),
]
MAX_VARIANCE = [(0.0, 0.0), (1.0, 0.0), (0.5, 0.25), (-1.0, 0.0)]

# TestLowRiskOverCommitment_Score: a pod without requests or limits scores MinNodeScore
SCORE_CASES = [
    dict(line=155, pod={"containers": []}, node={"cpu": "1000m", "memory": "1Gi"},
         metrics={0: [("CPU", "AVG", 20)]}, expected=[0]),
]

# computeRisk: SmoothingWindowSize 5, weights 0.5/0.5, node_A 4000m / 4Ki, CPU avg 80 std 0, Memory avg 25 std 0
NODE_A = {"cpu": "4000m", "memory": "4Ki"}
METRICS_A = [("CPU", "AVG", 80), ("CPU", "STD", 0), ("Memory", "AVG", 25), ("Memory", "STD", 0)]
NRLA_A1 = dict(req_cpu=2000, req_mem=2048, lim_cpu=3000, lim_mem=6144, req_minus_pod_cpu=1000, req_minus_pod_mem=0,
               lim_minus_pod_cpu=2000, lim_minus_pod_mem=0, cap_cpu=4000, cap_mem=4096)
NRLA_A2 = dict(req_cpu=4000, req_mem=1024, lim_cpu=5000, lim_mem=7168, req_minus_pod_cpu=3000, req_minus_pod_mem=512,
               lim_minus_pod_cpu=4000, lim_minus_pod_mem=6144, cap_cpu=4000, cap_mem=4096)
COMPUTE_RISK = [  # (name, metric type, nrla, want) — compared with == in the reference
    ("test-cpu-1", "CPU", NRLA_A1, 0.5),
    ("test-mem-1", "Memory", NRLA_A1, 0.25),
    ("test-cpu-2", "CPU", NRLA_A2, 1.0),
    ("test-mem-2", "Memory", NRLA_A2, 0.75),
]


def pod_with(overhead_milli, init_req, cont_req, init_lim=None, cont_lim=None):
    """getPodWithContainersAndOverhead (+ getPodWithLimits when limits are given): one init container, cpu in millicores,
    memory in bytes; container limits default to the requests (:625-627); getPodWithLimits is a no-op when the limit
    lists do not match the container count (:634-636)."""
    init = {"requests": {"cpu": f"{init_req[0]}m", "memory": init_req[1]}, "limits": {}}
    ctrs = [{"requests": {"cpu": f"{c}m", "memory": m}, "limits": {"cpu": f"{c}m", "memory": m}} for c, m in cont_req]
    if init_lim is not None and cont_lim is not None and len(cont_lim) == len(ctrs):
        init["limits"] = {"cpu": f"{init_lim[0]}m", "memory": init_lim[1]}
        for ctr, (c, m) in zip(ctrs, cont_lim):
            ctr["limits"] = {"cpu": f"{c}m", "memory": m}
    return {"containers": ctrs, "init_containers": [init], "overhead": {"cpu": f"{overhead_milli}m"}}


# TestGetResourceLimits: (pod, expected milliCPU, expected memory).  The second case reuses the first pod because the
# short CPU list makes getPodWithLimits return it unchanged.
_P = pod_with(10, (100, 512), [(1000, 2048), (500, 1024)], (2000, 4096), [(1000, 2048), (500, 1024)])
RESOURCE_LIMITS = [
    (_P, 2010, 4096),
    (_P, 2010, 4096),
    (pod_with(10, (100, 512), [(1000, 2048), (500, 1024)], (2000, 4096), [(1000, 4096), (2500, 2048)]), 3510, 6144),
    (pod_with(0, (100, 512), [(1000, 2048), (500, 1024), (0, 0)], (2000, 4096), [(0, 0), (0, 2048), (1000, 0)]), 2000, 4096),
]

# TestGetNodeRequestsAndLimits
_REQ = [(1000, 512), (500, 1024)]
POD = pod_with(0, (100, 2048), _REQ, (500, 2048), [(1500, 1024), (500, 1024)])          # also pod1, pod2
POD3 = pod_with(0, (100, 2048), _REQ, (500, 2048), [(1000, 1024), (1000, 2048)])
POD4 = pod_with(0, (100, 2048), _REQ, (500, 2048), [(1000, 1024), (400, 2048)])
TEST_NODE = {"cpu": "8000m", "memory": "6Ki"}
LOW_NODE = {"cpu": "1600m", "memory": "6Ki"}
NODE_REQUESTS_LIMITS = [
    dict(name="test-0", on_node=[POD, POD], node=TEST_NODE, pod=POD,
         want=dict(req_cpu=4500, req_mem=6144, lim_cpu=6000, lim_mem=6144, req_minus_pod_cpu=3000, req_minus_pod_mem=4096,
                   lim_minus_pod_cpu=4000, lim_minus_pod_mem=4096, cap_cpu=8000, cap_mem=6144)),
    dict(name="test-1", on_node=[POD3], node=TEST_NODE, pod=POD,
         want=dict(req_cpu=3000, req_mem=4096, lim_cpu=4000, lim_mem=5120, req_minus_pod_cpu=1500, req_minus_pod_mem=2048,
                   lim_minus_pod_cpu=2000, lim_minus_pod_mem=3072, cap_cpu=8000, cap_mem=6144)),
    # low-capacity node: requests 1500m / 2048, limits 2000m / 2048 (init container's 2048 memory dominates both)
    dict(name="test-2", on_node=[], node=LOW_NODE, pod=POD,
         want=dict(req_cpu=1500, req_mem=2048, lim_cpu=2000, lim_mem=2048, req_minus_pod_cpu=0, req_minus_pod_mem=0,
                   lim_minus_pod_cpu=0, lim_minus_pod_mem=0, cap_cpu=1600, cap_mem=6144)),
    # requests above limits: limits are raised to the requests (1500m; memory limit 3072)
    dict(name="test-3", on_node=[], node=LOW_NODE, pod=POD4,
         want=dict(req_cpu=1500, req_mem=2048, lim_cpu=1500, lim_mem=3072, req_minus_pod_cpu=0, req_minus_pod_mem=0,
                   lim_minus_pod_cpu=0, lim_minus_pod_mem=0, cap_cpu=1600, cap_mem=6144)),
]


def _pod(rc, rm, lc, lm):
    return {"containers": [{"requests": {"cpu": f"{rc}m", "memory": rm}, "limits": {"cpu": f"{lc}m", "memory": lm}}]}


# the same two fixtures rebuilt as pods, so that they can run through Score(): pods already on node_A whose sums are the
# *MinusPod fields, and a pending pod that brings NodeRequest / NodeLimit to the fixture's totals.
# rank = 1 - max(riskCPU, riskMemory) (lowriskovercommitment.go:165), score = round(100 * rank)
COMPUTE_RISK_AS_PODS = [
    dict(name="nrla_A1", on_node=[_pod(1000, 0, 2000, 0)], pod=_pod(1000, 2048, 1000, 6144), risk=(0.5, 0.25), score=50),
    dict(name="nrla_A2", on_node=[_pod(3000, 512, 4000, 6144)], pod=_pod(1000, 512, 1000, 1024), risk=(1.0, 0.75), score=0),
]
