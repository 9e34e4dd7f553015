"""bench.py reports roofline.traffic / valu_busy_frac from profiles/rNN/<workload>_traffic.json only while the kernels it runs are the
kernels that were profiled; the stamp is build.device_code_hash — the gfx950 machine code of the workload's kernel translation
units.  Checked here: the hash is a function of the compiled code only (same objects -> same hash, another unit -> another
hash), every committed profile of the latest round carries a stamp, and a stamp that does not match yields no counters."""
import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


@pytest.fixture(scope="module")
def bench_mod():
    import scheduler_plugins_amd as spx
    spx.lib()  # the library and its objects come from `python __graft_entry__.py build`; nothing here builds them
    import bench
    return bench


def test_device_code_hash_is_per_translation_unit(bench_mod):
    from scheduler_plugins_amd import build as B
    a = B.device_code_hash(["kernels_trimaran.hip"])
    assert a == B.device_code_hash(["kernels_trimaran.hip"]) and len(a) == 16
    assert a != B.device_code_hash(["kernels_network.hip"])
    assert B.device_code_hash(["kernels_trimaran.hip", "kernels_network.hip"]) == B.device_code_hash(["kernels_network.hip", "kernels_trimaran.hip"])
    with pytest.raises(OSError):
        B.device_code_hash(["no_such_unit.hip"])
    assert bench_mod.kernel_source_hash(("alloc", "tlp")) == a  # config #2's kernels live in that one unit


def test_committed_profiles_are_stamped_and_a_foreign_stamp_is_refused(bench_mod, tmp_path, monkeypatch):
    latest = sorted((ROOT / "profiles").glob("r*"))[-1]
    files = sorted(latest.glob("*_traffic.json"))
    assert files
    for f in files:
        d = json.loads(f.read_text())
        assert isinstance(d.get("kernel_source_hash"), str) and len(d["kernel_source_hash"]) == 16, f.name
        assert d["workload"] in bench_mod.WORKLOADS
    # a profile whose stamp is not this build's: no traffic, no VALU figure, the file named as stale
    d = json.loads((latest / "config2_traffic.json").read_text())
    d["kernel_source_hash"] = "0" * 16
    fake = tmp_path / "profiles" / "r99"
    fake.mkdir(parents=True)
    (fake / "config2_traffic.json").write_text(json.dumps(d))
    monkeypatch.setattr(bench_mod, "ROOT", tmp_path)
    got = bench_mod.profile_counters("config2", ("alloc", "tlp"))
    assert got["traffic"] is None and "stale_profile" in got and "valu_busy_frac" not in got


def test_latest_profiles_describe_the_kernels_in_the_tree(bench_mod):
    """Every profile of the latest round was taken with the machine code this tree builds: a kernel edit after the profiles were
    collected fails here (round-5 review: four workloads kept round-4 stamps and DESIGN printed their times as "unchanged") — re-collect
    with tools/prof_all.sh + tools/collect_profiles.py, or move the stale entry out of the latest round's directory."""
    latest = sorted((ROOT / "profiles").glob("r*"))[-1]
    stale = {}
    for f in sorted(latest.glob("*_traffic.json")):
        d = json.loads(f.read_text())
        want = bench_mod.kernel_source_hash(bench_mod.WORKLOADS[d["workload"]]["plugins"])
        if d["kernel_source_hash"] != want:
            stale[f.name] = (d["kernel_source_hash"], want)
    assert not stale, stale
