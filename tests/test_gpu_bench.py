"""bench.py's own code paths on the GPU box: the single-device line, and the multi-device mode (`--gpus N`, one host process
driving N devices through spx_multi) — the path the driver's scaling run takes.  On a one-GPU box `--devices 0,0 --transport copy`
puts two ranks on device 0 (same code: sharding, per-rank threads, spx_multi_mark / spx_multi_marked_ms timing, gather); with two or
more GPUs visible the same runs over RCCL."""
import ctypes as C
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from helpers import ALLOCATABLE, TLP
from scheduler_plugins_amd import synth
from scheduler_plugins_amd.engine import mask_of
from scheduler_plugins_amd.multi import PEER_COPY, MultiEngine

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def n_gpus():
    n = C.c_int(0)
    hip = C.CDLL("libamdhip64.so")
    return n.value if hip.hipGetDeviceCount(C.byref(n)) == 0 else 0


def run_bench(*args):
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def check_line(d, n_gpus_want, scaling):
    assert d["metric"] == "pod_x_node_filter_score_evals_per_sec" and d["unit"] == "evals/s"
    assert d["n_gpus"] == n_gpus_want and d["scaling"] == scaling
    assert d["ms_per_step"] > 0 and d["value"] > 0 and d["roofline"]["kernel_ms"] > 0
    assert 0 < d["roofline"]["frac"] <= 1.0
    assert d["roofline"]["kernel_ms"] <= d["ms_per_step"] * 1.5 + 0.05  # slowest rank's HIP-event time vs the wall clock around all ranks
    for key in ("gather", "topological_sort", "full_cycle"):
        if key in d:
            assert "error" not in d[key], d[key]


def test_single_device_line_config1(gpu_required):
    """BASELINE.json configs[0]: Allocatable (LeastAllocated), 100 nodes x 1k pods, through the C ABI"""
    d = run_bench("--workload", "config1", "--steps", "5", "--warmup", "2", "--cpu-budget", "1")
    check_line(d, 1, "weak")
    assert d["config"]["n_nodes"] == 100 and d["config"]["n_pods_per_step"] == 1000 and d["config"]["plugins"] == ["alloc"]
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"


@pytest.mark.parametrize("workload,scaling,extra", [
    ("small", "weak", []),
    ("small", "weak", ["--gather", "table"]),
    ("small_net", "strong", ["--gather", "table"]),
    ("small_full", "strong", []),
])
def test_multi_device_mode(gpu_required, workload, scaling, extra):
    d = run_bench("--workload", workload, "--devices", "0,0", "--transport", "copy", "--steps", "3", "--warmup", "1", "--cpu-budget", "0", *extra)
    check_line(d, 2, scaling)
    assert d["config"]["host"].startswith("one process driving 2 devices")
    g = d["gather"]
    assert g["best_ms"] > 0 and g["bytes_per_rank"] > 0 and g["step_plus_gather_ms"] > 0
    if "--gather" in extra:
        assert g["table_ms"] > 0 and g["table_bytes"] > 0
    if workload == "small_net":
        assert d["topological_sort"]["queue_sort_ms"] > 0 and d["topological_sort"]["n_keys"] == 8000
    # strong scaling shards ONE batch: the job's pods per step do not grow with the device count
    assert d["config"]["n_pods_per_step"] == (8000 if scaling == "strong" else 2 * 4000)
    assert d["config"]["n_pods_slowest_rank"] == 4000


@pytest.mark.parametrize("workload,gather", [("small_net", "table"), ("small_full", "best"), ("small_full_ragged", "best")])
def test_multi_device_mode_eight_ranks(gpu_required, workload, gather):
    """`--gpus 8` as one process sees it — eight ranks' engines, host threads, shards of 1 000 (the ragged batch: seven of 1 024 and one of
    1 023) and the exchange — with all eight on device 0 over peer copies: config #4's and config #5's plugin sets at plumbing size, so
    that the first run on eight devices is not the first run of this code with eight ranks"""
    d = run_bench("--workload", workload, "--devices", "0,0,0,0,0,0,0,0", "--transport", "copy", "--steps", "2", "--warmup", "1", "--cpu-budget", "0", "--gather", gather)
    check_line(d, 8, "strong")
    assert d["config"]["host"].startswith("one process driving 8 devices")
    total = 8191 if workload == "small_full_ragged" else 8000
    assert d["config"]["n_pods_per_step"] == total and d["config"]["n_pods_slowest_rank"] == -(-total // 8)
    assert d["gather"]["best_ms"] > 0 and d["collective"]["rccl_ranks"] == 0  # (peer copies: no RCCL communicator)
    if gather == "table":
        assert d["gather"]["table_bytes"] > 0


@pytest.mark.skipif(n_gpus() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("workload", ["small", "small_full"])
def test_multi_device_mode_rccl(gpu_required, workload):
    d = run_bench("--workload", workload, "--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-budget", "0", "--gather", "table")
    check_line(d, 2, "weak" if workload == "small" else "strong")
    assert d["gather"]["transport"] == "rccl"


def test_multi_mark_brackets_the_steps(gpu_required, hdr):
    """spx_multi_mark / spx_multi_marked_ms: per-rank HIP-event time between two marks — what `--gpus N` reports as kernel time"""
    snap = synth.trimaran_snapshot(hdr, 2000, 6000, seed=synth.SEED)
    with MultiEngine([0, 0], PEER_COPY) as m:
        m.load_trimaran_objects(snap["nodes"], snap["rc"], snap["pods"], snap["metrics"], snap.get("assigned"))
        mask = mask_of(ALLOCATABLE, TLP)
        with pytest.raises(Exception):
            m.marked_ms()  # no marks yet
        m.eval(mask)
        m.sync()
        m.mark(0)
        for _ in range(5):
            m.eval(mask)
        m.mark(1)
        m.sync()
        mx, per = m.marked_ms()
        assert len(per) == 2 and all(p > 0 for p in per) and mx == max(per)
        one = m.last_ms()[0]  # the last step alone
        assert mx >= one * 0.9  # five steps cannot be shorter than (nearly) one
        m.mark(0)
        m.mark(1)
        m.sync()
        assert m.marked_ms()[0] < mx  # marks move: an empty region is shorter than five steps


@pytest.mark.parametrize("workload,scaling,gather", [("small", "weak", "best"), ("small_net", "strong", "table"), ("small_full", "strong", "best")])
def test_ranks_mode_two_processes(gpu_required, workload, scaling, gather):
    """bench.py as the driver launches it for N > 1: `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`, one process
    per rank — sharding by rank, the barrier / max-over-ranks timing, the all-gather of the decisions (and of one table).  On a one-GPU
    box both ranks sit on device 0 and the collectives run over gloo (RCCL refuses two ranks on one device); with two GPUs visible
    the same test runs over RCCL.  No N > 1 hardware figure comes out of this: it is the code path, not the number."""
    two = n_gpus() >= 2
    extra = [] if two else ["--rank-devices", "0,0", "--dist-backend", "gloo"]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29613",
           str(ROOT / "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "3", "--warmup", "1", "--cpu-budget", "0", "--gather", gather, *extra]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]   # rank 0 prints, nobody else
    d = json.loads(lines[0])
    check_line(d, 2, scaling)
    assert d["config"]["host"].startswith("2 processes")
    assert d["gather"]["best_ms"] > 0 and d["gather"]["backend"] == ("nccl" if two else "gloo")
    if gather == "table":
        assert d["gather"]["table_bytes"] > 0


@pytest.mark.parametrize("ranks,port", [(2, 29615), (3, 29617)])
def test_ranks_mode_ragged_strong_shards_assemble(gpu_required, ranks, port):
    """a strong-scaling batch no rank count divides (8 191 pods over 2 and over 3 ranks: ceil(P/G) rows per rank, the last shard short)
    through bench.py's ranks mode; rank 0 evaluates the whole batch in one engine as well and every field of every all-gathered
    decision must equal it (round-4 review: two partition rules in one code base mis-assembled such shapes silently)"""
    have = n_gpus() >= ranks
    extra = [] if have else ["--rank-devices", ",".join(["0"] * ranks), "--dist-backend", "gloo"]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(ROOT / "bench.py"), "--gpus", str(ranks), "--workload", "small_full_ragged", "--steps", "2", "--warmup", "1", "--cpu-budget", "0", "--gather", "best",
           "--verify-gather", *extra]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][0])
    check_line(d, ranks, "strong")
    assert d["config"]["n_pods_per_step"] == 8191 and d["config"]["n_pods_slowest_rank"] == -(-8191 // ranks)
    assert d["gather"]["verified_rows"] == 8191 and d["gather"]["mismatches"] == 0, d["gather"]
