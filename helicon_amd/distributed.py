"""Multi-GPU sweep: one process per GPU, contiguous shards of the flat candidate index, one
all-gather of float32 scores (RCCL over xGMI when the backend is "nccl"; gloo on CPU tests).

The reference has no distributed code (its candidates are thread-pool tasks,
src/helicon/webApps/denovo3D/app.py:2473-2476); the partition follows SURVEY.md section 8e:
rank k owns ``[k*per, (k+1)*per)`` with ``per = ceil(G/W)`` rounded up to whole twists, short shards
are padded with NaN, every rank ends with all ``S x G`` scores and takes the arg-max locally (lowest
index on ties), so there is exactly one collective per sweep and no data-path exchange.

``ShardedSweep`` keeps everything a sweep needs between calls — the shard's parameters on the device,
the NaN-padded send buffer the library writes its scores straight into (``hh_sweep_device_strided``),
the receive buffer of the all-gather and the arg-max indices — so a step allocates nothing and never
synchronises the host: sweep kernels, the collective and the arg-max kernel are queued back to back.
"""
from __future__ import annotations

import numpy as np

from .grid import CandidateGrid, shard_bounds

__all__ = ["ShardedSweep", "gather_scores", "sweep_distributed", "shard_params", "harmless_rise",
           "assemble_scores", "best_from_blocks"]


def harmless_rise(grid: CandidateGrid) -> float:
    """Rise given to the pairs the reference's driver skips (they still occupy a slot; their scores
    are discarded): the smallest valid rise, so the slot changes neither the list's run structure
    nor the lattice size the library plans for."""
    ok = grid.params[grid.valid, 1]
    return float(ok.min()) if len(ok) else 1.0


def shard_params(params: np.ndarray, rank: int, world: int, align: int = 1):
    lo, hi, per = shard_bounds(len(params), rank, world, align)
    return params[lo:hi], lo, hi, per


def assemble_scores(blocks: np.ndarray, n_total: int) -> np.ndarray:
    """[W, S, per] rank-major all-gather output -> [S, G] in flat candidate order (pads dropped)."""
    w, s, per = blocks.shape
    return np.ascontiguousarray(blocks.transpose(1, 0, 2).reshape(s, w * per)[:, :n_total])


def best_from_blocks(values: np.ndarray, index: np.ndarray, per: int) -> np.ndarray:
    """Combine per-(rank, segment) arg-max results into the global one: ``values[W, S]`` are the block
    maxima (NaN = the block has no valid score), ``index[W, S]`` their positions inside the block.
    np.argmax's rule: the lowest flat index among equal maxima; NaN never wins; nothing valid -> 0."""
    w, s = values.shape
    out = np.zeros(s, dtype=np.int64)
    for k in range(s):
        best, have = 0.0, False
        for r in range(w):  # ascending rank = ascending flat index: strict '>' keeps the first maximum
            v = values[r, k]
            if v != v:
                continue
            if not have or v > best:
                best, have = v, True
                out[k] = r * per + int(index[r, k])
    return out


def gather_scores(local, n_total: int, per_rank: int, group=None):
    """``local``: torch tensor [S, n_local] float32 on this rank (CUDA for nccl, CPU for gloo).
    Returns [S, n_total] on every rank.  (Allocating form, kept for one-off calls; ``ShardedSweep``
    is the steady-state path.)"""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    s = local.shape[0]
    pad = torch.full((s, per_rank), float("nan"), dtype=torch.float32, device=local.device)
    pad[:, : local.shape[1]] = local
    out = torch.empty((world * s, per_rank), dtype=torch.float32, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)  # rank-major concatenation
    return out.view(world, s, per_rank).permute(1, 0, 2).reshape(s, world * per_rank)[:, :n_total].contiguous()


class ShardedSweep:
    """This rank's share of one candidate list, with every buffer of the sweep + all-gather + arg-max
    step allocated once.

    engine : a configured ``SweepEngine`` (geometry and reference set); its stream is bound to torch's
             current stream of the device so the collective is ordered behind the sweep without events.
    params : the full [G, 4] list (identical on every rank).
    align  : shard granularity in candidates (the number of rises: shards are whole twists).
    device : torch device of the buffers (default: the engine's GPU).  With a ``gloo`` group the
             collective is staged through host memory.
    """

    def __init__(self, engine, params: np.ndarray, align: int = 1, group=None, device=None):
        import torch
        import torch.distributed as dist

        self.engine, self.group = engine, group
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.backend = self.dist.get_backend(group) if self.dist else None
        self.n_total = len(params)
        self.lo, self.hi, self.per = shard_bounds(self.n_total, self.rank, self.world, align)
        self.n_local = self.hi - self.lo
        self.h_params = np.ascontiguousarray(params[self.lo:self.hi], dtype=np.float64)
        self.device = torch.device("cuda", engine.device) if device is None else torch.device(device)
        self.n_seg = max(1, int(engine.n_segments))
        s, per, w = self.n_seg, max(self.per, 1), self.world
        self.d_params = torch.from_numpy(self.h_params if self.n_local else np.zeros((1, 4))).to(self.device)
        self.send = torch.full((s, per), float("nan"), dtype=torch.float32, device=self.device)
        self.staged = self.device.type == "cuda" and self.backend == "gloo"
        if w > 1:
            cdev = "cpu" if self.staged else self.device
            self.recv = torch.empty((w, s, per), dtype=torch.float32, device=cdev)
            self.send_c = torch.empty((s, per), dtype=torch.float32, device="cpu") if self.staged else self.send
        else:
            self.recv = self.send.view(1, s, per)
        self.on_device = self.recv.device.type == "cuda"
        self.host_scores = self.host_index = None
        self.d_index = torch.zeros((w * s,), dtype=torch.int64, device=self.recv.device)
        if self.device.type == "cuda":
            engine.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def sweep(self):
        """Queue this shard's sweep; its scores land in the padded send buffer."""
        if self.n_local:
            self.engine.sweep_device(self.d_params.data_ptr(), self.n_local, self.send.data_ptr(),
                                     host_params=self.h_params, ld_scores=self.send.shape[1])

    def gather(self):
        """Queue the one collective of the sweep (nothing to do for a single rank)."""
        if self.world == 1:
            return
        if self.staged:
            self.send_c.copy_(self.send)  # synchronising D2H: the gloo rehearsal only
        w, s, per = self.recv.shape
        self.dist.all_gather_into_tensor(self.recv.view(w * s, per), self.send_c, group=self.group)

    def argmax(self):
        """Queue the per-(rank, segment) arg-max of the gathered blocks (device kernel; host loop under gloo)."""
        w, s, per = self.recv.shape
        if self.on_device:
            self.engine.argmax_device(self.recv.data_ptr(), w * s, per, per, d_index=self.d_index.data_ptr())
        else:
            from . import _lib
            import ctypes as C

            L = _lib.lib()
            idx = C.c_int64(0)
            flat = self.recv.view(w * s, per)
            for r in range(w * s):
                L.hh_argmax(C.cast(flat[r].data_ptr(), C.POINTER(C.c_float)), per, C.byref(idx))
                self.d_index[r] = idx.value

    def results_to_host(self):
        """Queue the copy of the gathered scores and the arg-max indices into pinned host buffers (SURVEY.md section 8d
        counts the result's device-to-host transfer inside the metric).  Asynchronous: complete after the stream is
        synchronised; ``host_scores`` / ``host_index`` then hold the step's results."""
        if self.recv.device.type != "cuda":
            self.host_scores, self.host_index = self.recv, self.d_index
            return
        if getattr(self, "host_scores", None) is None:
            import torch

            self.host_scores = torch.empty(self.recv.shape, dtype=self.recv.dtype, pin_memory=True)
            self.host_index = torch.empty(self.d_index.shape, dtype=self.d_index.dtype, pin_memory=True)
        self.host_scores.copy_(self.recv, non_blocking=True)
        self.host_index.copy_(self.d_index, non_blocking=True)

    def step(self, results_to_host=False):
        self.sweep()
        self.gather()
        self.argmax()
        if results_to_host:
            self.results_to_host()

    # -- results (these synchronise) ---------------------------------------------------------
    def scores(self) -> np.ndarray:
        """[S, G] float32 on the host, flat candidate order."""
        return assemble_scores(self.recv.cpu().numpy(), self.n_total)

    def best_index(self) -> np.ndarray:
        """[S] flat candidate index of each segment's arg-max (from the last ``argmax()``)."""
        w, s, per = self.recv.shape
        idx = self.d_index.cpu().numpy().reshape(w, s)
        blocks = self.recv.reshape(w * s, per)
        vals = blocks[np.arange(w * s), self.d_index.to(blocks.device)].cpu().numpy().reshape(w, s)
        return best_from_blocks(vals, idx, per)


def sweep_distributed(engine, grid: CandidateGrid, group=None, return_best=False):
    """Score ``grid`` with this rank's engine (geometry and reference already set), all-gather,
    return scores [S, G] as a NumPy array on every rank (and the per-segment arg-max indices)."""
    params = grid.params.copy()
    params[~grid.valid, 1] = harmless_rise(grid)
    sh = ShardedSweep(engine, params, align=len(grid.rises), group=group)
    sh.step()
    full = sh.scores()
    if return_best:
        masked = full.copy()
        masked[:, ~grid.valid] = -np.inf
        return full, np.array([int(np.argmax(np.where(np.isnan(m), -np.inf, m))) for m in masked])
    return full
