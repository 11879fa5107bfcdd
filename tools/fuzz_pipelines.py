#!/usr/bin/env python3
"""Randomised agreement campaign between the pipelines (run on the GPU box): for seeded random geometries and
twist-major grids, whatever pipeline the library picks must reproduce the general (transform) pipeline.
    python tools/fuzz_pipelines.py [cases] [seed] [side]     (side: force one image side, e.g. 1024)
Rises are drawn so that the per-group row counts sit near their integer boundaries as often as not."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import helicon_amd as H  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
force_n = int(sys.argv[3]) if len(sys.argv) > 3 else 0
picked = {"fused": 0, "run_tables": 0, "transform": 0}
worst = 0.0
engines = {}
for case in range(cases):
    n = int(rng.choice([32, 64, 128, 256, 512], p=[0.2, 0.33, 0.3, 0.14, 0.03]))
    n = force_n or n
    apix = float(rng.choice([1.0, 1.37, 2.0, 3.3]))
    br = float(rng.uniform(0.8, 4.5) * apix) if case % 5 else float(rng.uniform(5.0, 7.5) * apix)
    d = float(rng.uniform(0.1, 0.97) * (0.99 * n * apix - br))
    dy = float(rng.choice([0.0, rng.uniform(-0.2, 0.2) * n * apix]))
    rot = float(rng.choice([0.0, rng.uniform(-180, 180)]))
    csym = int(rng.integers(1, 8))
    # window of four columns: (3 + 2 rpx) apix; choose rises around span / integer
    sigma2 = br * br / np.log(2.0)
    rpx = max(1, int(np.ceil(np.sqrt(sigma2 * 24 * np.log(2.0)) / apix)))  # default tail_bits
    span4 = (3 + 2 * rpx) * apix
    if case % 2:
        rise0 = span4 / int(rng.integers(1, 9)) * float(rng.choice([0.9999, 1.0, 1.0001, 0.97, 1.03]))
    else:
        rise0 = float(rng.choice([rng.uniform(1.0, 30.0) * apix, rng.uniform(0.3, 1.0) * apix, n * apix * 0.3]))
    rise0 = min(rise0, 0.45 * n * apix)
    n_rises = int(rng.choice([8, 9, 16, 33, 70, 250, 701], p=[0.2, 0.15, 0.2, 0.15, 0.15, 0.1, 0.05]))
    rises = rise0 * (1.0 + float(rng.choice([1e-4, 3e-3, 0.02])) * np.arange(n_rises))
    # (now and then many runs: the launch then spans several rounds of resident workgroups and the schedule cuts the
    # last runs finer than the first ones)
    twists = np.round(rng.uniform(-170, 170, int(rng.integers(1, 5)) if case % 7 else int(rng.integers(20, 120))), 3)
    units = None
    if case % 4 == 0:
        k = int(rng.integers(2, 4))
        units = np.stack([rng.uniform(0.2, 0.5, k) * d, rng.uniform(-3, 3, k), rng.uniform(-6, 6, k) * apix], axis=1)
    params = np.array([[tw, rs, csym, rot] for tw in twists for rs in rises])
    if case % 6 == 1 and len(params) > 24:  # ragged list: starts and / or stops inside a run
        cut = params[int(rng.integers(0, n_rises)): len(params) - int(rng.integers(0, n_rises))]
        params = cut if len(cut) >= 8 else params
    kind = case % 3
    if kind == 0:
        mask = rng.random((n, n)) < 0.4
    elif kind == 1:
        mask = H.radial_band_mask(n, n)
    else:  # resolution-limited band: whole ky blocks carry no weight and are skipped
        r_hi = float(rng.uniform(0.15, 0.5) * n)
        mask = H.radial_band_mask(n, n, float(rng.uniform(0.0, 3.0)), r_hi)
        if not mask.any():
            mask = H.radial_band_mask(n, n)
    mb = int(rng.choice([0, 16, 50]))
    eng = engines.get((n, mb))
    if eng is None:
        eng = engines[(n, mb)] = H.SweepEngine(n, max_batch=mb)
    eng.set_table_path(2)
    tail_bits = int(rng.choice([0, 0, 0, 12, 30]))
    eng.set_geometry(apix=apix, helical_diameter=d, ball_radius=br, dy=dy, units=units, tail_bits=tail_bits)
    img = eng.simulate(float(twists[0]), float(rises[n_rises // 2]), csym, rot)
    img = (img + rng.normal(0, 0.3 * img.std() + 1e-3, img.shape)).astype(np.float32)
    segs = img if case % 7 else np.stack([img, img[::-1].copy()])
    eng.set_reference(segs, mask, log=bool(case % 2))
    got = eng.sweep(params)
    picked[eng.last_first_pass] += 1
    eng.set_table_path(0)
    want = eng.sweep(params)
    err = float(np.abs(got - want).max())
    worst = max(worst, err)
    if err > 5e-5:
        print(f"MISMATCH case {case}: n={n} apix={apix} br={br:.3f} d={d:.2f} rise0={rise0:.6f} x{n_rises} c={csym} rot={rot:.2f} "
              f"dy={dy:.2f} units={None if units is None else len(units)} mb={mb} err={err:.3e}", flush=True)
    if case % 50 == 49:
        print(f"{case + 1} cases, worst |dscore| {worst:.2e}, pipelines {picked}", flush=True)
print(f"done: {cases} cases, worst |dscore| {worst:.2e}, pipelines {picked}")
