"""Multi-GPU host logic: independent registrations shard across ranks with NO data-path
collective; the only exchange is one all-gather of each rank's n x 12-float result block (pose + stats).
Backend-agnostic: "nccl" (= RCCL over xGMI) on the GPU node, "gloo" in the CPU tests."""
from __future__ import annotations

import numpy as np

RESULT_SIZE = 12


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition: rank r owns [lo, hi).  Sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_results(T: np.ndarray, stats: list[dict]) -> np.ndarray:
    """The 12-float record liblisreg keeps on the device: T[6], iters, deltaR, deltaT, degenerate, n_corr, status."""
    out = np.zeros((len(stats), RESULT_SIZE), np.float32)
    out[:, :6] = np.asarray(T, np.float32).reshape(-1, 6)
    for i, s in enumerate(stats):
        out[i, 6:] = [s["iters"], s["deltaR"], s["deltaT"], s["degenerate"], s["n_corr_last"], s["status"]]
    return out


def gather_results(local_block, n_items: int, world: int, group=None):
    """All-gather ragged per-rank result blocks (torch tensors [n_local, 12]) -> [n_items, 12] in item order.
    Ranks pad to the largest shard so one fixed-size all_gather suffices (latency-bound, 48 B per item)."""
    import torch
    import torch.distributed as dist
    n_max = -(-n_items // world)
    pad = torch.zeros((n_max, RESULT_SIZE), dtype=torch.float32, device=local_block.device)
    pad[: local_block.shape[0]] = local_block
    out = torch.empty((world, n_max, RESULT_SIZE), dtype=torch.float32, device=local_block.device)
    dist.all_gather_into_tensor(out.view(-1), pad.view(-1), group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_items, r, world)
        parts.append(out[r, : hi - lo])
    return torch.cat(parts, 0)
