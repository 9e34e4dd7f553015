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
