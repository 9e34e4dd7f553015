"""Multi-GPU sharding of the pods x nodes evaluation.

Pod rows are the independent unit given a frozen snapshot: rank g owns rows [g*ceil(P/G), (g+1)*ceil(P/G)) ∩ [0, P), node-side tables
are replicated, and the evaluation itself needs no collective.  What a consumer wants back is either
  * the per-pod decision (best node, weighted score, tie count, feasible count): 20 bytes per pod — one small
    RCCL all-gather, or
  * the full uint8 tables on every GPU: an all-gather of the slabs, xGMI-bound ((G-1)/G x table / ~1 TB/s).
Both are plain torch.distributed collectives on the engine's own device buffers; backend "nccl" is RCCL on ROCm,
"gloo" is used by the CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_rows(n_pods: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range of `rank`: ceil(n_pods / world) rows per rank, the last ranks short or empty — THE partition rule of
    this code base (spx_multi_shard in csrc/spx_multi.hip and bench.py's ranks mode use the same one: global row g lives on rank
    g // ceil(P/G) at local row g % ceil(P/G), which is what lets equal slabs be all-gathered in place)."""
    per = -(-n_pods // world) if world > 0 else 0
    begin = min(n_pods, rank * per)
    return begin, min(n_pods, begin + per)


def shard_sizes(n_pods: int, world: int) -> List[int]:
    return [shard_rows(n_pods, world, r)[1] - shard_rows(n_pods, world, r)[0] for r in range(world)]


def gather_best(dist, device, node: np.ndarray, score: np.ndarray, ties: np.ndarray, feasible: np.ndarray, n_pods: int):
    """All-gathers the per-pod decisions of every rank's shard into full-length arrays (ragged shards are padded
    to the longest one, so a single all_gather_into_tensor moves everything)."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = shard_sizes(n_pods, world)
    longest = max(sizes)
    packed = torch.zeros((longest, 4), dtype=torch.int64, device=device)
    mine = sizes[rank]
    local = np.stack([node.astype(np.int64), score.astype(np.int64), ties.astype(np.int64), feasible.astype(np.int64)], axis=1)
    packed[:mine] = torch.from_numpy(local[:mine]).to(device)
    out = torch.empty((world * longest, 4), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, packed)
    out = out.cpu().numpy().reshape(world, longest, 4)
    full = np.concatenate([out[r, : sizes[r]] for r in range(world)], axis=0)
    return full[:, 0].astype(np.int32), full[:, 1], full[:, 2].astype(np.int32), full[:, 3].astype(np.int32)


def gather_table(dist, slab):
    """All-gathers equal-sized uint8 slabs ([rows_per_rank, row_stride] torch tensors on the collective's device)
    into the full [world*rows_per_rank, row_stride] table on every rank."""
    import torch

    world = dist.get_world_size()
    out = torch.empty((world * slab.shape[0], slab.shape[1]), dtype=slab.dtype, device=slab.device)
    dist.all_gather_into_tensor(out, slab.contiguous())
    return out
