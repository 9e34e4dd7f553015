import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import scheduler_plugins_amd as spx
        e = spx.Engine(0)
        e.close()
        return True
    except Exception:
        return False


@pytest.fixture(scope="session")
def hdr():
    import scheduler_plugins_amd as spx
    return spx.header()


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def gpu_required():
    """GPU tests must fail loudly (not skip) when the HIP extension cannot run on a GPU box."""
    import scheduler_plugins_amd as spx
    spx.lib()  # ImportError if libspx.so is missing
    return spx
