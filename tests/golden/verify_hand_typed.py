#!/usr/bin/env python3
"""Checks hand-typed golden tables against the reference's Go test tables, read with goparse (no Go toolchain here, nothing is
executed).  Runs only where /root/reference is mounted; tests/test_golden_hand_typed.py calls it and is skipped elsewhere.
Covered so far: allocatable.py (pkg/noderesources/allocatable_test.go:114-238) — every case's pod, node sizes, resource weights,
mode, expected list and line; the other hand-typed files are listed in README.md as not yet machine-checked."""
from __future__ import annotations

import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from goparse import Call, Ident, line_of, parse_literal_after  # noqa: E402

REF = Path("/root/reference")


def check_allocatable() -> int:
    import allocatable as A
    src = (REF / "pkg/noderesources/allocatable_test.go").read_text()
    p = src.index("func TestNodeResourcesAllocatable")
    score = {"fwk.MinNodeScore": 0, "fwk.MaxNodeScore": 100}
    sets = {"defaultResourceAllocatableSet": A.DEFAULT, "cpuResourceAllocatableSet": A.CPU_HEAVY}
    pods = {"cpuAndMemory": A.CPU_AND_MEMORY, "bigCpu": A.BIG_CPU}
    checked = 0
    hand = {c["line"]: c for c in A.CASES}
    bad = {c["line"]: c for c in A.INVALID}
    for t in parse_literal_after(src[p:], "tests := "):
        name = t["name"]
        # the hand-typed tables cite the line of a case's first field: the one after its opening brace
        at = src.index('"' + name + '"', p)
        start = src.rfind("\n\t\t{\n", 0, at) + 3
        line = src.count("\n", 0, start) + 2
        args = t["args"]
        res = args["Resources"]
        if isinstance(res, Ident):
            weights = sets[res.name]
        else:
            weights = {e["Name"]: e["Weight"] for e in res}  # (literal lists name the resources with plain strings)
        if "wantErr" in t:
            c = bad[line + 1]  # (these two open with a comment line)
            assert c["resources"] == weights and c["name"] == name, (line, weights, c)
            checked += 1
            continue
        c = hand[line]
        # (names are abbreviated in the hand-typed table — and two of the reference's own names contradict the case's Mode, :183-196 — the line ties them)
        pod = t["pod"]
        want_pod = pods[pod.name] if isinstance(pod, Ident) else A.NO_RESOURCES
        assert c["pod"] == want_pod, (line, pod)
        nodes = [(n.args[1], n.args[2]) for n in t["nodeInfos"]]
        nodes = [tuple(eval_const(v) for v in n) for n in nodes]
        assert c["nodes"] == nodes, (line, nodes, c["nodes"])
        assert c["resources"] == weights, (line, weights)
        assert c["mode"] == {"modeLeast": "Least", "modeMost": "Most"}[args["Mode"].name], line
        exp = [eval_const(e["Score"], score) for e in t["expectedList"]]
        assert c["expected"] == exp, (line, exp, c["expected"])
        checked += 1
    assert checked == len(A.CASES) + len(A.INVALID), (checked, len(A.CASES), len(A.INVALID))
    return checked


def eval_const(v, names=None):
    """integer expressions as the test writes them: 1000 * 1024 * 1024, (fwk.MinNodeScore + fwk.MaxNodeScore) / 2"""
    if isinstance(v, int):
        return v
    if isinstance(v, Ident):
        return names[v.name]
    if isinstance(v, Call) and v.fn in ("op*", "op+", "op/"):
        a = [eval_const(x, names) for x in v.args]
        out = a[0]
        for x in a[1:]:
            out = out * x if v.fn == "op*" else (out + x if v.fn == "op+" else out // x)
        return out
    raise ValueError(repr(v))


if __name__ == "__main__":
    print("allocatable.py:", check_allocatable(), "cases agree with allocatable_test.go")
