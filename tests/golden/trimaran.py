"""Known answers of the reference's trimaran tests, as data.

TLP: pkg/trimaran/targetloadpacking/targetloadpacking_test.go:148-238 (TestTargetLoadPackingScoring).
  The test builds the plugin with TargetUtilization=40, DefaultRequestsMultiplier="1.5" and NO
  DefaultRequests, so requestsMilliCores = 0 (targetloadpacking.go:79).  Node: cpu 1000m, memory 1Gi
  (Capacity == Allocatable, st.MakeNode().Capacity()).
LVRB: .../loadvariationriskbalancing/analysis_test.go:38-157 (TestComputeScore),
  .../loadvariationriskbalancing_test.go:152-328 (TestScore; margin 1, sensitivity 1 :330-334),
  pkg/trimaran/resourcestats_test.go:77-161 (TestCreateResourceStats), :259-372 (TestGetMuSigma).
"""
MEGA = 1024 * 1024
NODE = {"cpu": "1000m", "memory": "1Gi"}

TLP_PARAMS = dict(target_utilization=40, default_requests_milli=0, requests_multiplier=1.5)


def _pod_overhead(overhead, *requests):  # getPodWithContainersAndOverhead targetloadpacking_test.go:436-450
    return {"containers": [{"requests": {"cpu": f"{r}m"}, "limits": {"cpu": f"{r}m"}} for r in requests],
            "overhead": {"cpu": f"{overhead}m"}}


TLP_CASES = [
    dict(name="new node", line=156, pod={"containers": []}, metrics={0: [("CPU", "Latest", 0)]}, expected=[40]),
    dict(name="hot node", line=182, pod={"containers": []}, metrics={0: [("CPU", "Latest", 50)]}, expected=[33]),
    dict(name="excess utilization returns min score", line=209, pod=_pod_overhead(0, 1000),
         metrics={0: [("CPU", "Latest", 30)]}, expected=[0]),
    dict(name="404 resp from watcher", line=227, pod={"containers": []}, metrics=None, expected=[0]),
]

# (margin, sensitivity, Capacity, Req, UsedAvg, UsedStdev) -> int64(math.Round(computeScore))
COMPUTE_SCORE = [
    ("valid data", 1, 1, 100, 10, 40, 36, 57),
    ("zero capacity", 1, 2, 0, 10, 40, 36, 0),
    ("negative usedAvg", 1, 2, 100, 10, -40, 36, 65),
    ("large usedAvg", 1, 2, 100, 10, 200, 36, 20),
    ("negative usedStdev", 1, 2, 100, 10, 40, -36, 75),
    ("large usedStdev", 1, 2, 100, 10, 40, 120, 25),
    ("large usedAvg (dup)", 1, 2, 100, 10, 200, 36, 20),
    ("negative margin", -1, 1, 100, 10, 40, 36, 75),
    ("negative sensitivity", 1, -1, 100, 10, 40, 36, 57),
    ("zero sensitivity", 1, 0, 100, 10, 40, 36, 75),
]

# (Capacity, Req, UsedAvg, UsedStdev) -> (mu, sigma), compared with == in the reference
MU_SIGMA = [
    ("proper arguments", 1000, 100, 400, 360, 0.5, 0.36),
    ("zero arguments", 0, 0, 0, 0, 0.0, 0.0),
    ("large used", 1000, 100, 1400, 300, 1.0, 0.3),
    ("large deviation", 1000, 100, 400, 1600, 0.5, 1.0),
    ("large arguments", 1000, 0, 1400, 1600, 1.0, 1.0),
    ("negative used", 1000, 0, -100, 200, 0.0, 0.2),
    ("negative deviation", 1000, 400, 0, -200, 0.4, 0.0),
]


def _lv_pod(cpu_reqs, mem_reqs):  # getPodWithContainersAndOverhead(0,0,0,[]cpu,[]mem) loadvariationriskbalancing_test.go:415-442
    return {"init_containers": [{"requests": {"cpu": "0m", "memory": "0"}}],
            "containers": [{"requests": {"cpu": f"{c}m", "memory": str(m)}, "limits": {"cpu": f"{c}m", "memory": str(m)}}
                           for c, m in zip(cpu_reqs, mem_reqs)],
            "overhead": {"cpu": "0m"}}


LVRB_CASES = [
    dict(name="new node", line=160, pod={"containers": []}, metrics={0: [("CPU", "AVG", 50)]}, expected=[75]),
    dict(name="hot node", line=186, pod={"containers": []}, metrics={0: [("CPU", "AVG", 100)]}, expected=[50]),
    dict(name="average and stDev metrics", line=213, pod=_lv_pod([200], [256 * MEGA]),
         metrics={0: [("CPU", "AVG", 30), ("CPU", "STD", 16)]}, expected=[67]),
    dict(name="CPU and Memory metrics", line=244, pod=_lv_pod([100], [512 * MEGA]),
         metrics={0: [("CPU", "AVG", 40), ("CPU", "STD", 16), ("Memory", "AVG", 50), ("Memory", "STD", 10)]},
         expected=[45]),
    dict(name="pick worst case: CPU or Memory", line=285, pod=_lv_pod([100], [512 * MEGA]),
         metrics={0: [("CPU", "AVG", 80), ("CPU", "STD", 20), ("Memory", "AVG", 25), ("Memory", "STD", 15)]},
         expected=[45]),
    dict(name="404 resp from watcher", line=326, pod={"containers": []}, metrics=None, expected=[0]),
]

# TestCreateResourceStats resourcestats_test.go:36-161: metrics list, podRequest {100m, 1Mi}
STATS_METRICS = [("CPU", "", 40), ("CPU", "AVG", 40), ("CPU", "STD", 36), ("Memory", "AVG", 20), ("Memory", "STD", 10)]
STATS_EXPECT = {
    "cpu": dict(capacity=1000.0, req=100.0, used_avg=400.0, used_stdev=360.0),
    "memory": dict(capacity=1024.0, req=1.0, used_avg=204.8, used_stdev=102.4),
}
