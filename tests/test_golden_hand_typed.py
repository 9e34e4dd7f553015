"""Hand-typed golden tables re-read from the reference's Go sources (tests/golden/verify_hand_typed.py).  Only where /root/reference
is mounted (the build container); the GPU box and CI without the reference skip it."""
import sys
from pathlib import Path

import pytest

REF = Path("/root/reference/pkg/noderesources/allocatable_test.go")


@pytest.mark.skipif(not REF.exists(), reason="/root/reference is not mounted here")
def test_allocatable_table_agrees_with_the_go_source():
    sys.path.insert(0, str(Path(__file__).parent / "golden"))
    import verify_hand_typed
    assert verify_hand_typed.check_allocatable() == 14


@pytest.mark.skipif(not REF.exists(), reason="/root/reference is not mounted here")
def test_trimaran_lroc_peaks_tables_agree_with_the_go_sources():
    """trimaran.py COMPUTE_SCORE + MU_SIGMA (17 rows), lroc.py's Beta-distribution tables (10 rows), network.py's Score and Filter cases (11), peaks.py's power model and
    NormalizeScore cases — re-read from analysis_test.go, resourcestats_test.go, beta_test.go, peaks_test.go"""
    sys.path.insert(0, str(Path(__file__).parent / "golden"))
    import verify_hand_typed
    assert verify_hand_typed.check_trimaran() == 17
    assert verify_hand_typed.check_trimaran_score_cases() == 10
    assert verify_hand_typed.check_lroc() == 10
    assert verify_hand_typed.check_lroc_compute_risk() == 7
    assert verify_hand_typed.check_network() == 11
    assert verify_hand_typed.check_nrt_helpers() == 44
    assert verify_hand_typed.check_nrt_helpers_pods() == 22
    assert verify_hand_typed.check_nrt_helpers_numa_lists() == 11
    assert verify_hand_typed.check_peaks() == 10


@pytest.mark.skipif(not REF.exists(), reason="/root/reference is not mounted here")
def test_round5_tables_agree_with_the_go_sources():
    """what round 4's review listed as still hand-typed only and round 5 re-read: trimaran.py's STATS_* (TestCreateResourceStats), lroc.py's
    GetResourceLimits / GetNodeRequestsAndLimits tables and Score case (the straight-line fixture code of resourcestats_test.go read by a
    small statement reader), nrt_helpers.py's ONLY_NON_NUMA_ZONES and the two OVER_RESERVE flows (straight-line tests, regular expressions)"""
    sys.path.insert(0, str(Path(__file__).parent / "golden"))
    import verify_hand_typed
    assert verify_hand_typed.check_trimaran_stats() == 10
    assert verify_hand_typed.check_lroc_resource_tables() == 9
    assert verify_hand_typed.check_nrt_helpers_zones_and_over_reserve() == 4


@pytest.mark.skipif(not REF.exists(), reason="/root/reference is not mounted here")
def test_preemption_and_integration_tables_agree_with_the_go_sources():
    """the last three hand-typed files: nrt_preemption.py (TestGetNRTPostPodsEviction: fixtures, every case's victims / placement / error / expected zones),
    nrt_preemption_flow.py (TestFilter_PreemptionFlow: the seven straight-line sub-tests read as statements) and integration.py (the five integration tests'
    metrics literals, node and pod quantities, expected placements)"""
    sys.path.insert(0, str(Path(__file__).parent / "golden"))
    import verify_hand_typed
    assert verify_hand_typed.check_nrt_preemption() == 10
    assert verify_hand_typed.check_nrt_preemption_flow() == 10
    assert verify_hand_typed.check_integration() == 6
