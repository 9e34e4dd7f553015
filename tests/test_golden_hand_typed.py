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
