"""N > 1 path on CPU: world_size-2 gloo run of the score all-gather and the shard arithmetic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helicon_amd.distributed import gather_scores, shard_params
from helicon_amd.grid import build_grid


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, n_seg, aligned, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grid = build_grid(np.arange(1.0, 1.0 + 0.1 * 13, 0.1)[:13], np.arange(4.0, 4.0 + 0.5 * (n_total // 13 + 1), 0.5)[: n_total // 13 + 1],
                          (1,), tube_length=1e9)
        params = grid.params[:n_total]
        # aligned: shards start on a twist (what sweep_distributed does), else the plain ceil(G/W) blocks
        mine, lo, hi, per = shard_params(params, rank, world, align=len(grid.rises) if aligned else 1)
        assert not aligned or lo % len(grid.rises) == 0
        # stand-in scores: a deterministic function of the candidate, so the gather can be checked
        base = torch.from_numpy((mine[:, 0] * 1000 + mine[:, 1]).astype(np.float32))
        local = torch.stack([base + 0.25 * s for s in range(n_seg)]) if hi > lo else torch.empty((n_seg, 0))
        full = gather_scores(local, n_total, per)
        expect = (params[:, 0] * 1000 + params[:, 1]).astype(np.float32)
        ok = full.shape == (n_seg, n_total)
        for s in range(n_seg):
            ok = ok and np.array_equal(full[s].numpy(), expect + np.float32(0.25 * s))
        q.put((rank, bool(ok), int(torch.argmax(full[0]))))
    except Exception as e:  # report instead of leaving the parent to time out
        q.put((rank, False, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total,n_seg,aligned", [(101, 1, False), (64, 3, False), (1, 1, False), (101, 2, True)])
def test_allgather_of_scores_world2(n_total, n_seg, aligned):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, n_seg, aligned, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert len({am for _, _, am in res}) == 1  # every rank agrees on the arg-max
