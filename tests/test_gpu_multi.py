"""Two ranks over RCCL when two devices are visible (the round-end driver's 8-GPU node; a one-GPU box runs the same child
script as a world of one): every rank is a fresh process started before anything touched the GPU, takes its whole-twist
shard of ONE candidate list, and must end up with the one-rank sweep's scores and arg-max — through the C ABI's own
collective (hh_comm_* / hh_allgather) and through helicon_amd.distributed.ShardedSweep over torch.distributed.
SURVEY.md section 8(e); the reference has no distributed code (its candidates are pool tasks, app.py:2473-2476)."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _device_count():
    """Counted in a child (torch.cuda.device_count does not initialise the GPU here, but keep this process out of it)."""
    out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
    return int(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else 0


def test_shards_over_rccl_equal_the_one_rank_sweep(tmp_path):
    world = min(2, _device_count())
    assert world >= 1, "no GPU visible"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "multi_rank_child.py"), str(r), str(world), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]

    import helicon_amd as H
    from helicon_amd.distributed import assemble_scores
    from oracle import path_b as O

    n, apix = 64, 2.0
    d, br = 0.4 * n * apix, 2 * apix
    clean = O.simulate_helical_projection(1, 29.0, 10.0, 1, d, br, 0, 0, n, n, apix)
    imgs = np.stack([(clean + np.random.default_rng(s).normal(0, 0.5 * clean.std(), clean.shape)).astype(np.float32) for s in range(2)])
    grid = H.build_grid(np.arange(25.0, 33.5, 1.0), np.arange(8.0, 12.5, 0.5), (1,), tube_length=n * apix)
    with H.SweepEngine(n) as eng:
        eng.set_geometry(apix=apix, helical_diameter=d, ball_radius=br)
        eng.set_reference(imgs)
        ref = eng.sweep(grid.params)
    for r in range(world):
        blocks = np.load(tmp_path / f"abi_{r}.npy")                      # [world][segments][per]
        np.testing.assert_array_equal(assemble_scores(blocks, len(grid)), ref)
        np.testing.assert_array_equal(np.load(tmp_path / f"torch_{r}.npy").reshape(ref.shape), ref)
        tb = np.load(tmp_path / f"torch_best_{r}.npy")
        assert [int(v) for v in tb] == [int(np.nanargmax(ref[s])) for s in range(2)]
    if world == 1:
        pytest.skip("one device visible: the child ran as a world of one (the two-rank run needs two GPUs)")
