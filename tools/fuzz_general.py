#!/usr/bin/env python3
"""Randomised campaign for the general-size path (run on the GPU box): random (ny, nx) whose row length factors into
2, 3, 5, 7, 11, 13, random geometry, masks and grids with long, short and single runs — the device's scores against the
NumPy oracle on a sample of every grid.
    python tools/fuzz_general.py [cases] [seed]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import helicon_amd as H  # noqa: E402
from oracle import path_b as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def smooth_side(lo, hi):
    while True:
        n = int(rng.integers(lo, hi))
        m = n
        for p in (2, 3, 5, 7, 11, 13):
            while m % p == 0:
                m //= p
        if m == 1:
            return n


worst = 0.0
for case in range(cases):
    ny, nx = int(rng.integers(16, 90)), smooth_side(16, 130)
    if case % 5 == 0:
        ny, nx = smooth_side(100, 260), smooth_side(100, 330)
    apix = float(rng.choice([1.0, 1.5, 2.0, 3.0]))
    br = float(rng.uniform(1.2, 3.0) * apix)
    d = float(rng.uniform(0.2, 0.9) * (0.99 * ny * apix - br))
    csym = int(rng.integers(1, 5))
    n_tw = int(rng.choice([1, 2, 7, 40]))
    n_rs = int(rng.choice([1, 8, 17, 60, 300]))
    if n_tw * n_rs > 4000:
        n_tw = max(1, 4000 // n_rs)
    twists = np.round(rng.uniform(-170, 170, n_tw), 3)
    rise0 = float(rng.uniform(2.0, 12.0) * apix)
    rises = rise0 * (1.0 + float(rng.choice([1e-3, 0.01])) * np.arange(n_rs))
    params = np.array([[tw, rs, csym, 0.0] for tw in twists for rs in rises])
    if case % 4 == 1 and len(params) > 12:   # a ragged list
        params = params[int(rng.integers(0, 5)): len(params) - int(rng.integers(0, 5))]
    mask = H.radial_band_mask(ny, nx) if case % 3 else (rng.random((ny, nx)) < 0.5)
    log = bool(case % 2)
    eng = H.SweepEngine((ny, nx))
    try:
        eng.set_geometry(apix=apix, helical_diameter=d, ball_radius=br)
    except AssertionError:
        continue
    img = eng.simulate(float(twists[0]), float(rises[len(rises) // 2]), csym)
    img = (img + rng.normal(0, 0.3 * img.std() + 1e-3, img.shape)).astype(np.float32)
    eng.set_reference(img[None], mask, log=log)
    try:
        got = eng.sweep(params)[0]
    except ValueError as e:    # (rise too small for the general path's 32-row limit and the like: loud refusals)
        print(f"case {case}: refused: {e}")
        continue
    pick = np.unique(np.r_[0, len(params) - 1, rng.integers(0, len(params), 10)])
    ref = O.sweep_cpu(img, params[pick, :3], mask, apix=apix, helical_diameter=d, ball_radius=br, log=log)
    err = float(np.abs(got[pick] - ref).max())
    worst = max(worst, err)
    if err > 2e-4 or not np.isfinite(got).all():
        print(f"MISMATCH case {case}: {ny}x{nx} apix={apix} br={br:.2f} d={d:.1f} csym={csym} grid {n_tw}x{n_rs} rise0={rise0:.3f} "
              f"log={log} err={err:.2e}", flush=True)
    if case % 20 == 19:
        print(f"{case + 1} cases, worst |dscore| {worst:.2e}", flush=True)
print(f"done: {cases} cases, worst |dscore| {worst:.2e}")
