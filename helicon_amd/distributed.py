"""Multi-GPU sweep: one process per GPU, contiguous shards of the flat candidate index, one
all-gather of float32 scores (RCCL over xGMI when the backend is "nccl"; gloo on CPU tests).

The reference has no distributed code (its candidates are thread-pool tasks,
src/helicon/webApps/denovo3D/app.py:2473-2476); the partition follows SURVEY.md section 8e:
rank k owns ``[k*ceil(G/W), (k+1)*ceil(G/W))``, short shards are padded with NaN, every rank
ends with all ``S x G`` scores and takes the arg-max locally (lowest index on ties), so there is
exactly one collective per sweep and no data-path exchange.
"""
from __future__ import annotations

import numpy as np

from .grid import CandidateGrid, shard_bounds

__all__ = ["gather_scores", "sweep_distributed", "shard_params", "harmless_rise"]


def harmless_rise(grid: CandidateGrid) -> float:
    """Rise given to the pairs the reference's driver skips (they still occupy a slot; their scores
    are discarded): the smallest valid rise, so the slot changes neither the list's run structure
    nor the lattice size the library plans for."""
    ok = grid.params[grid.valid, 1]
    return float(ok.min()) if len(ok) else 1.0


def shard_params(params: np.ndarray, rank: int, world: int, align: int = 1):
    lo, hi, per = shard_bounds(len(params), rank, world, align)
    return params[lo:hi], lo, hi, per


def gather_scores(local, n_total: int, per_rank: int, group=None):
    """``local``: torch tensor [S, n_local] float32 on this rank (CUDA for nccl, CPU for gloo).
    Returns [S, n_total] on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    s = local.shape[0]
    pad = torch.full((s, per_rank), float("nan"), dtype=torch.float32, device=local.device)
    pad[:, : local.shape[1]] = local
    out = torch.empty((world * s, per_rank), dtype=torch.float32, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)  # rank-major concatenation
    return out.view(world, s, per_rank).permute(1, 0, 2).reshape(s, world * per_rank)[:, :n_total].contiguous()


def sweep_distributed(engine, grid: CandidateGrid, group=None):
    """Score ``grid`` with this rank's engine (geometry and reference already set), all-gather,
    return scores [S, G] as a NumPy array on every rank."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    params = grid.params.copy()
    params[~grid.valid, 1] = harmless_rise(grid)
    mine, lo, hi, per = shard_params(params, rank, world, align=len(grid.rises))
    dev = torch.device("cuda", engine.device)
    d_params = torch.from_numpy(np.ascontiguousarray(mine)).to(dev)
    d_scores = torch.empty((engine.n_segments, max(hi - lo, 1)), dtype=torch.float32, device=dev)
    engine.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    if hi > lo:
        engine.sweep_device(d_params.data_ptr(), hi - lo, d_scores.data_ptr(), host_params=mine)
    full = gather_scores(d_scores[:, : hi - lo], len(params), per, group)
    return full.cpu().numpy()
