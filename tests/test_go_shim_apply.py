"""shim/go/apply_shim.py still finds every method it replaces in the surveyed reference checkout, and the edited files stay
well-formed (braces balance, the new body is in, the old per-(pod,node) computation is out).  The shim itself cannot be compiled
here (no Go toolchain): this keeps the edit list honest against the reference's file layout."""
import importlib.util
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")


@pytest.mark.skipif(not REF.exists(), reason="the reference checkout only exists in the build container")
def test_every_edit_applies(tmp_path):
    spec = importlib.util.spec_from_file_location("apply_shim", ROOT / "shim" / "go" / "apply_shim.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    done = mod.apply(REF, tmp_path)
    assert sum(n for _, n in done) == 9 and len(done) == 6
    for rel, _ in done:
        new = (tmp_path / rel).read_text()
        assert new.count("{") == new.count("}")
        assert ".spx." in new or "ts.order[" in new
    tlp = (tmp_path / "pkg/trimaran/targetloadpacking/targetloadpacking.go").read_text()
    assert "pl.spx.ScoreRow(pod, spx.PluginTLP)" in tlp and "PredictUtilisation(&container)" not in tlp.split("func (pl *TargetLoadPacking) Score")[1].split("\nfunc ")[0]


def test_shim_sources_name_only_exported_entry_points():
    """every C.spx_* the Go files call is declared in include/spx.h"""
    import re
    hdr = (ROOT / "include" / "spx.h").read_text()
    for f in (ROOT / "shim" / "go" / "pkg" / "spx").glob("*.go"):
        for name in set(re.findall(r"C\.(spx_[a-z_0-9]+)\(", f.read_text())):
            assert re.search(rf"\b{name}\(", hdr), (f.name, name)
        for const in set(re.findall(r"C\.(SPX_[A-Z_0-9]+)", f.read_text())):
            assert re.search(rf"#define {const}\b", hdr), (f.name, const)
