"""The C++ host-side mirror of the plugin interface (scheduler-plugins_amd/host/plugins.hpp) driven with
upstream's call pattern: 16 concurrent readers per pod (tests/cpp/harness.cc)."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_cpp_harness_16_concurrent_readers(gpu_required):
    exe = ROOT / "tests" / "cpp" / "_build" / "harness"
    assert exe.exists(), "run `python __graft_entry__.py build` first"
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "harness ok" in r.stdout
