#!/usr/bin/env python3
"""BASELINE configuration C5 (64 segments x 512^2 against one 200 x 100 grid) alone, as bench.py's C5 leg runs it,
for rocprofv3 or for tuning: argv = [repetitions, segments, outer radius of the mask in bins (default: the whole disc)]."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import helicon_amd as H  # noqa: E402
from helicon_amd.distributed import ShardedSweep  # noqa: E402
from helicon_amd.grid import build_grid, sweep_axis  # noqa: E402

if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    segments = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = torch.device("cuda:0")
    eng = H.SweepEngine(512, device=0)
    eng.set_geometry(apix=1.0, helical_diameter=0.4 * 512, ball_radius=2.0)
    clean = eng.simulate(1.20, 4.75, 1)
    imgs = np.stack([(clean + np.random.default_rng(s).normal(0, 0.5 * clean.std(), clean.shape)).astype(np.float32)
                     for s in range(segments)])
    r_hi = float(sys.argv[3]) if len(sys.argv) > 3 else None
    eng.set_reference(imgs, H.radial_band_mask(512, 512, 2.0, r_hi) if r_hi else None, log=True)
    twists, rises = sweep_axis(0.02, 4.00, 0.02), sweep_axis(4.25, 5.24, 0.01)
    grid = build_grid(twists, rises, (1,), tube_length=512.0)
    sh = ShardedSweep(eng, grid.params, align=len(rises), device=dev)
    sh.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sh.step(results_to_host=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    best = sh.best_index()
    truth = int(np.argmin(np.abs(grid.params[:, 0] - 1.20) + np.abs(grid.params[:, 1] - 4.75)))
    print(f"C5: {len(grid)} candidates x {segments} segments, {dt * 1e3:.3f} ms per step = {len(grid) * segments / dt / 1e6:.1f} M scores/s; "
          f"{int(sum(int(b) == truth for b in best))} of {segments} segments at the truth; memory {eng.memory_bytes()}", flush=True)
