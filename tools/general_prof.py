#!/usr/bin/env python3
"""The general-size sweep of bench.py's general_400 / general_200 legs, alone, for rocprofv3 `--kernel-trace --stats`:
argv = [side (400 | 200 | any) or ny,nx, repetitions, segments]; prints candidates/s of the timed steps."""
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
    shape = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "400").split(",")]
    ny, side = (shape[0], shape[-1])
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    segments = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    dev = torch.device("cuda:0")
    eng = H.SweepEngine((ny, side), device=0)
    eng.set_geometry(apix=1.0, helical_diameter=0.4 * ny, ball_radius=2.0)
    clean = eng.simulate(1.20, 4.75, 1)
    imgs = np.stack([(clean + np.random.default_rng(sg).normal(0, 0.5 * clean.std(), clean.shape)).astype(np.float32) for sg in range(segments)])
    eng.set_reference(imgs if segments > 1 else imgs[0], None, log=True)
    tw, rs = sweep_axis(0.01, 4.00, 0.01), sweep_axis(4.000, 5.245, 0.005)
    twists = tw[70:170] if side >= 300 else tw[20:220]
    grid = build_grid(twists, rs, (1,), tube_length=float(side))
    sh = ShardedSweep(eng, grid.params, align=len(rs), device=dev)
    sh.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sh.step(results_to_host=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    bests = sh.best_index()
    best = int(bests[0])
    print(f"{ny} x {side}: {len(grid)} candidates x {segments} segment(s), {dt * 1e3:.3f} ms per step = {len(grid) / dt / 1e6:.3f} M candidates/s; "
          f"best (twist, rise) = ({grid.params[best, 0]:.2f}, {grid.params[best, 1]:.3f}); segments agreeing with the first: "
          f"{int(sum(int(b) == best for b in bests))}", flush=True)
