"""The N>1 path on CPU: two gloo ranks shard the pod rows, evaluate their shard (here: with the CPU oracle standing
in for the GPU engine, which cannot run without a GPU), all-gather decisions and tables, and must reproduce the
unsharded result."""
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch, torch.distributed as dist
    ROOT = sys.argv[1]
    sys.path[:0] = [ROOT, ROOT + "/oracle", ROOT + "/tests"]
    import scheduler_plugins_amd as spx
    from scheduler_plugins_amd import synth, shard
    from helpers import tlp_params
    import pyoracle

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    hdr = spx.header()
    N, P = 200, int(os.environ.get('SPX_TEST_PODS', '101'))   # ragged: 51 + 50 rows (101, 2 ranks); 34 + 34 + 33 / 4 + 4 + 2 with 3
    snap = synth.trimaran_snapshot(hdr, N, P, seed=21)
    osnap = pyoracle.Snapshot(snap["nodes"], snap["pods"], metrics=snap["metrics"], assigned=snap["assigned"], tlp_params=tlp_params(hdr))
    b, e = shard.shard_rows(P, world, rank)
    local, _ = osnap.score_rows(1, b, e)                      # this rank's TLP rows
    best_node = local.argmax(axis=1).astype(np.int32)
    best_score = local.max(axis=1)
    ties = (local == best_score[:, None]).sum(axis=1).astype(np.int32)
    feas = np.full(e - b, N, np.int32)
    node, score, t, f = shard.gather_best(dist, torch.device("cpu"), best_node, best_score, ties, feas, P)
    full, _ = osnap.score_rows(1)                             # unsharded
    assert np.array_equal(node, full.argmax(axis=1)) and np.array_equal(score, full.max(axis=1))
    assert np.array_equal(t, (full == full.max(axis=1)[:, None]).sum(axis=1)) and (f == N).all()
    # table gather needs equal slabs: pad the shorter shard like the engine's row ranges would be padded
    rows = max(shard.shard_sizes(P, world))
    slab = torch.zeros((rows, N), dtype=torch.uint8)
    slab[: e - b] = torch.from_numpy(local.astype(np.uint8))
    table = shard.gather_table(dist, slab).numpy()
    sizes = shard.shard_sizes(P, world)
    got = np.concatenate([table[r * rows: r * rows + sizes[r]] for r in range(world)])
    assert np.array_equal(got, full.astype(np.uint8))
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_shard_rows_partition():
    """one partition rule everywhere: ceil(P/G) rows per rank (shard.shard_rows = spx_multi_shard = bench.py's ranks mode)"""
    sys.path.insert(0, str(ROOT))
    import ctypes as C

    import scheduler_plugins_amd as spx
    from scheduler_plugins_amd import shard
    for p, w in [(100000, 8), (101, 2), (7, 8), (1, 1), (0, 4), (10, 4), (8191, 2), (8191, 3), (500000, 8), (200000, 3)]:
        ranges = [shard.shard_rows(p, w, r) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == p
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        per = -(-p // w)
        assert all((b, e) == (min(p, r * per), min(p, (r + 1) * per)) for r, (b, e) in enumerate(ranges))
        assert sum(shard.shard_sizes(p, w)) == p
    assert [shard.shard_rows(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]   # not 3,3,2,2


@pytest.mark.parametrize("world,n_pods,port", [(2, 101, 29531), (3, 10, 29533)])
def test_gloo_sharded_eval_and_gather(tmp_path, world, n_pods, port):
    """2 ranks x 101 rows (51 + 50) and 3 ranks x 10 rows (4 + 4 + 2: the shape on which a balanced rule and the ceil rule differ)"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SPX_TEST_PODS=str(n_pods))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script), str(ROOT)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count("ok") >= world
