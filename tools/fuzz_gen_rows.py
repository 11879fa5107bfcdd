#!/usr/bin/env python3
"""Every row length the two-step row kernel is instantiated for (gen_rows.hip: nx = R1 R2, 109 pairs) against the NumPy
oracle on a small grid, and against the Stockham kernel where that one fits (run on the GPU box):
    python tools/fuzz_gen_rows.py [ny]"""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import helicon_amd as H  # noqa: E402
from helicon_amd import _lib  # noqa: E402
from helicon_amd.grid import build_grid  # noqa: E402
from oracle import path_b as O  # noqa: E402

ny = int(sys.argv[1]) if len(sys.argv) > 1 else 24
L = _lib.lib()
sizes = []
for nx in range(32, 1025):
    out = (C.c_int64 * 6)()
    assert L.hh_general_plan(nx, 2 * (nx // 8) + 1, 7, out) == 0
    if out[0] and not (nx & (nx - 1) == 0 and nx == ny):
        sizes.append((nx, int(out[0]), int(out[1])))
print(f"{len(sizes)} row lengths, ny = {ny}", flush=True)
worst, worst_vs = 0.0, 0.0
for nx, r1, r2 in sizes:
    apix, tw0, rs0 = 2.0, 31.0, 9.0
    d, br = 0.4 * ny * apix, 2 * apix
    clean = O.simulate_helical_projection(1, tw0, rs0, 1, d, br, 0, 0, ny, nx, apix)
    img = (clean + np.random.default_rng(nx).normal(0, 0.5 * clean.std(), clean.shape)).astype(np.float32)
    grid = build_grid(tw0 + np.array([-0.6, 0.0]), rs0 + np.array([-0.2, 0.0, 0.3]), (1,), tube_length=nx * apix)
    mask = O.radial_band_mask(ny, nx)
    with H.SweepEngine((ny, nx)) as eng:
        eng.set_geometry(apix=apix, helical_diameter=d, ball_radius=br)
        eng.set_reference(img, mask)
        got = eng.sweep(grid.params)[0]
        vs = None
        if nx <= 600:
            os.environ["HH_GEN_STOCKHAM"] = "1"
            try:
                vs = float(np.abs(eng.sweep(grid.params)[0] - got).max())
            finally:
                del os.environ["HH_GEN_STOCKHAM"]
    ref = O.sweep_cpu(img, grid.params[:, :3], mask, apix=apix, helical_diameter=d, ball_radius=br)
    err = float(np.abs(got - ref).max())
    worst = max(worst, err)
    if vs is not None:
        worst_vs = max(worst_vs, vs)
    flag = "" if err < 2e-5 and (vs is None or vs < 5e-6) else "   <-- CHECK"
    print(f"nx {nx:4d} = {r1:2d} x {r2:2d}: |score - oracle| {err:.2e}" + (f", |two-step - Stockham| {vs:.2e}" if vs is not None else "") + flag, flush=True)
print(f"done: {len(sizes)} sizes, worst against the oracle {worst:.2e}, against the Stockham kernel {worst_vs:.2e}")
