#!/usr/bin/env python3
"""Single-GPU timing + sanity of the BASELINE configurations other than the bench line (C2):
C3 (Csym 1..6), C4 (1024^2), C5 (64 segments, shared 20k grid).  Prints one line per config."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import helicon_amd as H  # noqa: E402


def make(n, n_seg, truth=(1.20, 4.75, 1), mask=None):
    eng = H.SweepEngine(n)
    apix = 1.0
    eng.set_geometry(apix=apix, helical_diameter=0.4 * eng.ny * apix, ball_radius=2 * apix)
    clean = eng.simulate(*truth)
    imgs = np.stack([(clean + np.random.default_rng(s).normal(0, 0.5 * clean.std(), clean.shape)).astype(np.float32)
                     for s in range(n_seg)])
    eng.set_reference(imgs, mask)
    return eng


def run(name, eng, grid, reps=2):
    for _ in range(2):  # untimed: buffers grow to the sweep's size on first use, clocks settle
        eng.sweep(grid.params)
    t0 = time.perf_counter()
    for _ in range(reps):
        sc = eng.sweep(grid.params)
    dt = (time.perf_counter() - t0) / reps
    best = [tuple(np.round(grid.params[int(np.argmax(s)), :3], 4)) for s in sc[:3]]
    print(f"{name}: {len(grid)} candidates x {sc.shape[0]} segment(s) in {dt * 1e3:.1f} ms = "
          f"{len(grid) / dt:,.0f} cand/s ({len(grid) * sc.shape[0] / dt:,.0f} scores/s); pipeline {eng.last_first_pass}; "
          f"best {best}", flush=True)


tw, rs = H.sweep_axis(0.01, 4.00, 0.01), H.sweep_axis(4.000, 5.245, 0.005)
which = sys.argv[1:] or ["C3", "C4", "C5", "LOWRES", "GEN", "SMALL", "SHARDS"]
if "C3" in which:
    run("C3 512^2, 400x250 grid x Csym 1..6", make(512, 1), H.build_grid(tw, rs, (1, 2, 3, 4, 5, 6), tube_length=512.0), reps=1)
if "C4" in which:
    run("C4 1024^2, 500x500 grid", make(1024, 1),
        H.build_grid(H.sweep_axis(0.01, 5.00, 0.01), H.sweep_axis(4.000, 6.495, 0.005), (1,), tube_length=1024.0), reps=1)
if "C5" in which:
    run("C5 64 segments x 512^2, 200x100 grid", make(512, 64),
        H.build_grid(H.sweep_axis(0.02, 4.00, 0.02), H.sweep_axis(4.25, 5.24, 0.01), (1,), tube_length=512.0))
if "LOWRES" in which:  # resolution-limited mask: only ky blocks below the cut-off are stored / read
    run("C2 grid, 512^2, mask 2 < r < 100 (13 of 32 ky blocks)", make(512, 1, mask=H.radial_band_mask(512, 512, 2, 100)),
        H.build_grid(tw, rs, (1,), tube_length=512.0))
if "GEN" in which:  # sizes that are not powers of two: the runtime-sized kernels (hh_create2)
    for shape, nt in (((200, 200), 200), ((400, 400), 100), ((300, 480), 100)):
        run(f"general {shape[0]}x{shape[1]}, {nt}x250 grid", make(shape, 1),
            H.build_grid(tw[:nt], rs, (1,), tube_length=float(shape[1])), reps=1)
if "SMALL" in which:  # the smaller power-of-two sides through the tuned kernels (configs[0] is a 256 x 256 image)
    for n in (256, 128):
        run(f"{n}^2, 400x250 grid", make(n, 1), H.build_grid(tw, rs, (1,), tube_length=float(n)), reps=3)
if "SHARDS" in which:  # what one rank of a strong-scaling run sweeps: whole twists of the C2 grid, 1/N of them
    eng = make(512, 1)
    for ranks in (1, 2, 4, 8):
        run(f"C2 shard of 1/{ranks} ({len(tw) // ranks} twists x 250 rises)", eng,
            H.build_grid(tw[: len(tw) // ranks], rs, (1,), tube_length=512.0), reps=5)
