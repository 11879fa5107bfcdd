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


# ------------------------------------------------------------------------------------------------------------
# ShardedSweep (persistent buffers, sweep straight into the padded send buffer, block-wise arg-max) with the
# stand-in engine of tests/fake_engine.py, world size 2 over gloo, against the single-process result
# ------------------------------------------------------------------------------------------------------------
def _sharded_worker(rank, world, port, n_tw, n_rs, n_seg, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helicon_amd.distributed import ShardedSweep
        from tests.fake_engine import FakeEngine, fake_scores

        grid = build_grid(1.0 + 0.05 * np.arange(n_tw), 4.5 + 0.02 * np.arange(n_rs), (1, 2), tube_length=1e9)
        eng = FakeEngine(64)
        eng.set_reference(np.zeros((n_seg, 64, 64), np.float32))
        sh = ShardedSweep(eng, grid.params, align=n_rs, device="cpu")
        assert sh.lo % n_rs == 0 and sh.send.shape == (n_seg, sh.per)
        ptrs = (sh.send.data_ptr(), sh.recv.data_ptr(), sh.d_params.data_ptr())
        for _ in range(3):  # steady state: the same buffers every step
            sh.step()
        assert ptrs == (sh.send.data_ptr(), sh.recv.data_ptr(), sh.d_params.data_ptr())
        full = sh.scores()
        expect = np.stack([fake_scores(grid.params, s) for s in range(n_seg)])
        ok = full.shape == expect.shape and np.array_equal(full, expect)
        best = sh.best_index()
        ok = ok and np.array_equal(best, np.argmax(expect, axis=1))
        q.put((rank, bool(ok), best.tolist()))
    except Exception as e:
        q.put((rank, False, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_tw,n_rs,n_seg", [(7, 9, 1), (8, 16, 3), (1, 5, 2)])
def test_sharded_sweep_world2_equals_single_process(n_tw, n_rs, n_seg):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, n_tw, n_rs, n_seg, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]


def test_block_argmax_combination_rule():
    from helicon_amd.distributed import assemble_scores, best_from_blocks

    nan = np.nan
    vals = np.array([[0.5, nan, 0.1], [0.5, nan, 0.3], [0.4, 0.2, nan]], dtype=np.float32)      # [W=3, S=3]
    idx = np.array([[3, 0, 1], [0, 0, 2], [1, 4, 0]])
    assert best_from_blocks(vals, idx, per=10).tolist() == [3, 24, 12]     # tie -> lowest flat index; NaN blocks skipped
    assert best_from_blocks(np.full((2, 1), nan, np.float32), np.zeros((2, 1), int), 5).tolist() == [0]
    blocks = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    full = assemble_scores(blocks, 7)
    assert full.shape == (3, 7) and full[1].tolist() == [4, 5, 6, 7, 16, 17, 18]


# ------------------------------------------------------------------------------------------------------------
# bench.py --gpus 2 started WITHOUT a launcher: it must spawn its own ranks, shard the one grid, and print one
# JSON line whose scores / arg-max equal the one-rank run's (stand-in engine, gloo, CPU)
# ------------------------------------------------------------------------------------------------------------
def _run_bench(tmp_path, gpus, extra=()):
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    dump = tmp_path / f"scores_{gpus}.npy"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PYTHONPATH"] = str(root) + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--side", "64",
           "--engine", "tests.fake_engine:FakeEngine", "--backend", "gloo", "--no-cpu-baseline", "--dump-scores", str(dump),
           *extra]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0]), np.load(dump)


def test_bench_self_launch_world2_matches_world1(tmp_path):
    one, s1 = _run_bench(tmp_path, 1)
    two, s2 = _run_bench(tmp_path, 2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert one["config"]["candidates_per_step"] == two["config"]["candidates_per_step"] == 100000
    assert two["config"]["candidates_per_rank"] == 50000
    np.testing.assert_array_equal(s1, s2)
    assert one["argmax"] == two["argmax"] and one["argmax"]["is_truth"]
    weak, sw = _run_bench(tmp_path, 2, ["--scaling", "weak"])
    assert weak["scaling"] == "weak" and weak["config"]["candidates_per_step"] == 200000 and sw.shape == (1, 200000)
    np.testing.assert_array_equal(sw[:, :100000], s1)


def test_bench_config_c5_world2_matches_world1(tmp_path):
    """`bench.py --config C5` (64 segments against one shared grid, per-segment arg-max) at reduced size: 3 segments,
    every 10th twist — two self-launched ranks over gloo give the one-rank scores and arg-max for every segment."""
    extra = ["--config", "C5", "--segments", "3", "--grid-stride", "10"]
    one, s1 = _run_bench(tmp_path, 1, extra)
    two, s2 = _run_bench(tmp_path, 2, extra)
    assert s1.shape == (3, 20 * 100) and one["config"]["segments"] == 3 and one["config"]["candidates_per_step"] == 2000
    assert two["config"]["candidates_per_rank"] == 1000
    np.testing.assert_array_equal(s1, s2)
    assert one["argmax"] == two["argmax"] and one["argmax"]["segments"] == 3
    c4, s4 = _run_bench(tmp_path, 2, ["--config", "C4", "--grid-stride", "50"])
    assert s4.shape == (1, 10 * 500) and c4["config"]["image"] == 64        # (--side 64 overrides the configuration's 1024)
    c3, s3 = _run_bench(tmp_path, 2, ["--config", "C3", "--grid-stride", "40"])
    assert s3.shape == (1, 6 * 10 * 250)
