"""One rank of the multi-GPU sweep, started as a FRESH process by tests/test_gpu_multi.py (before anything touched the GPU):
    python tests/multi_rank_child.py <rank> <world> <dir>
Part 1, no host framework in the data path: hh_comm_unique_id (rank 0, handed over through a file) / hh_comm_init / this rank's
whole-twist shard through hh_sweep_device_strided into the NaN-padded send buffer / hh_allgather / hh_argmax_device.
Part 2: helicon_amd.distributed.ShardedSweep over torch.distributed (backend nccl = RCCL).  Each rank writes what it gathered;
the parent compares with a one-rank sweep."""
import os
import sys
import time
from pathlib import Path

import numpy as np

rank, world, out = int(sys.argv[1]), int(sys.argv[2]), Path(sys.argv[3])
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import helicon_amd as H  # noqa: E402
from helicon_amd.distributed import ShardedSweep, shard_params  # noqa: E402
from oracle import path_b as O  # noqa: E402

n, apix = 64, 2.0
d, br = 0.4 * n * apix, 2 * apix
clean = O.simulate_helical_projection(1, 29.0, 10.0, 1, d, br, 0, 0, n, n, apix)
imgs = np.stack([(clean + np.random.default_rng(s).normal(0, 0.5 * clean.std(), clean.shape)).astype(np.float32) for s in range(2)])
grid = H.build_grid(np.arange(25.0, 33.5, 1.0), np.arange(8.0, 12.5, 0.5), (1,), tube_length=n * apix)
rises = 9

import torch  # noqa: E402

torch.cuda.set_device(rank)
with H.SweepEngine(n, device=rank) as eng:
    eng.set_geometry(apix=apix, helical_diameter=d, ball_radius=br)
    eng.set_reference(imgs)
    # ---- part 1: the C ABI's own collective
    idf = out / "unique_id.bin"
    if rank == 0:
        tmp = out / "unique_id.tmp"
        tmp.write_bytes(H.SweepEngine.comm_unique_id())
        tmp.rename(idf)
    t0 = time.time()
    while not idf.exists():
        if time.time() - t0 > 120:
            raise SystemExit("no unique id from rank 0")
        time.sleep(0.05)
    eng.comm_init(rank, world, idf.read_bytes())
    mine, lo, hi, per = shard_params(grid.params, rank, world, align=rises)
    dp = torch.from_numpy(np.ascontiguousarray(mine)).cuda(rank)
    send = torch.full((2, per), float("nan"), dtype=torch.float32, device=f"cuda:{rank}")
    recv = torch.zeros((world, 2, per), dtype=torch.float32, device=f"cuda:{rank}")
    torch.cuda.synchronize(rank)
    if len(mine):
        eng.sweep_device(dp.data_ptr(), len(mine), send.data_ptr(), host_params=mine, ld_scores=per)
    eng.allgather(send.data_ptr(), 2 * per, recv.data_ptr())
    best = eng.argmax_device(recv.data_ptr(), 2 * world, per, per)
    eng.synchronize()
    np.save(out / f"abi_{rank}.npy", recv.cpu().numpy())
    np.save(out / f"abi_best_{rank}.npy", np.asarray(best))
    eng.comm_destroy()
    # ---- part 2: ShardedSweep over torch.distributed / RCCL
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        sh = ShardedSweep(eng, grid.params, align=rises, device=torch.device("cuda", rank))
        sh.step(results_to_host=True)
        np.save(out / f"torch_{rank}.npy", np.asarray(sh.scores()))
        np.save(out / f"torch_best_{rank}.npy", np.asarray(sh.best_index()))
    finally:
        dist.destroy_process_group()
print("rank", rank, "of", world, "done", flush=True)
