"""Multi-GPU: independent windows shard across ranks with no data-path collective.

Window w goes to rank w mod G.  The only communication is one all-gather of fixed-size per-window result
records (RCCL over xGMI when the tensors live on GPUs; gloo on CPU in the tests).  Splitting ONE window over
several GPUs is deliberately not offered: it would need an all-reduce of the normal equations per LM iteration.
"""
from __future__ import annotations

import numpy as np

RECORD = 6  # window id, iterations, termination code, initial cost, final cost, final radius


def shard(n_windows: int, rank: int, world: int) -> list[int]:
    """Window ids owned by `rank` (static round-robin partition)."""
    return list(range(rank, n_windows, world))


def make_records(window_ids, summaries) -> np.ndarray:
    from .capi import TERMINATION
    inv = {v: k for k, v in TERMINATION.items()}
    rec = np.zeros((len(window_ids), RECORD))
    for i, (wid, s) in enumerate(zip(window_ids, summaries)):
        rec[i] = [wid, s["iterations"], inv.get(s["termination"], -1), s["initial_cost"], s["final_cost"], s["final_radius"]]
    return rec


def gather_records(local: np.ndarray, n_windows: int, device=None) -> np.ndarray:
    """All-gather the per-window records of every rank; returns (n_windows, RECORD) ordered by window id.
    Uses torch.distributed if initialised (backend nccl = RCCL on GPU tensors, gloo on CPU), else returns local."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        out = np.full((n_windows, RECORD), np.nan)
        out[local[:, 0].astype(int)] = local
        return out
    world = dist.get_world_size()
    per = (n_windows + world - 1) // world
    buf = torch.full((per, RECORD), float("nan"), dtype=torch.float64, device=device)
    if len(local):
        buf[: len(local)] = torch.as_tensor(local, dtype=torch.float64, device=device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    allrec = torch.cat(parts).cpu().numpy()
    allrec = allrec[~np.isnan(allrec[:, 0])]
    out = np.full((n_windows, RECORD), np.nan)
    out[allrec[:, 0].astype(int)] = allrec
    return out
