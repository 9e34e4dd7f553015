"""Known answers of the reference's Peaks tests, as data.

pkg/trimaran/peaks/peaks_test.go: power model fixture (:80-86), TestPeaksScore (:165-423; the expected raw score of the
first case is computed in the test itself as int64(getPowerJumpForUtilisation(0, 100, model) * math.Pow(10, 15)), :238-240),
TestPeaksNormalizeScore (:426-531)."""
import math

POWER_MODEL = {"k0": 471.7412504314313, "k1": -91.50493019588365, "k2": -0.07186049052516228}  # node-1
NODE = {"cpu": "1000m", "memory": "1Gi"}


def jump(x, p, m=POWER_MODEL):  # getPowerJumpForUtilisation peaks.go:186-188
    return m["k1"] * (math.exp(m["k2"] * p) - math.exp(m["k2"] * x))


SCORE_TO_USE = int(jump(0, 100) * 1e15)  # ~9.14e16; equal to Go's value up to the last digits of math.Exp

_REQ = {"containers": [{"requests": {"cpu": 1, "memory": 2}, "limits": {}}]}          # MakeResourceList().CPU(1).Mem(2)
_POD3 = {"containers": [], "overhead": {"cpu": 0}}                                     # Overhead = a zero cpu quantity (:219-222)
_POD4 = {"containers": [{"requests": {}, "limits": {"cpu": "2000m"}}]}                 # limits only (:223-231)
SCORE_CASES = [  # metrics: {node: [(type, operator, value)]} or None for "404 resp from watcher"
    dict(name="Pod with Requests", line=250, pod=_REQ, metrics={0: [("CPU", "Latest", 0)]}, expected=SCORE_TO_USE, exact=False),
    dict(name="No CPU metrics found", line=275, pod=_REQ, metrics={0: [("Memory", "Latest", 0)]}, expected=0, exact=True),
    dict(name="Pod with Overhead", line=301, pod=_POD3, metrics={0: [("CPU", "Latest", 0)]}, expected=0, exact=True),
    dict(name="Pod with above node resource capacity", line=328, pod=_POD4, metrics={0: [("CPU", "Latest", 100)]}, expected=0, exact=True),
    dict(name="No watcher response for node", line=354, pod=_POD4, metrics={}, expected=0, exact=True),
    dict(name="404 resp from watcher", line=370, pod={"containers": []}, metrics=None, expected=0, exact=True),
]
NORMALIZE_CASES = [  # (line, scores before, after)
    (478, [0, 100], [100, 0]),
    (487, [0, 0], [0, 0]),
    (496, [100, 100], [100, 100]),
]
