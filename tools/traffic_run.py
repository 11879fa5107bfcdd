#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc traffic passes: calibration launches (known byte counts in the
sweep's own access shapes) followed by one C2 sweep through each of the three pipelines.  Run it once per counter:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out_fetch -- python tools/traffic_run.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out_write -- python tools/traffic_run.py

then `python tools/traffic_parse.py out_fetch out_write` writes profiles/traffic.json."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import helicon_amd as H  # noqa: E402

CALIB_BYTES = 1 << 30  # 1 GiB: 4x the Infinity Cache

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
eng = H.SweepEngine(n)
for _ in range(3):
    eng.calibrate_traffic(0, CALIB_BYTES)
    eng.calibrate_traffic(1, CALIB_BYTES)
apix = 1.0
eng.set_geometry(apix=apix, helical_diameter=0.4 * n * apix, ball_radius=2 * apix)
clean = eng.simulate(1.2, 4.75, 1)
img = (clean + np.random.default_rng(0).normal(0, 0.5 * clean.std(), clean.shape)).astype(np.float32)
eng.set_reference(img)
nt = 400 if n <= 512 else 100  # (the 1024 side is four times the work per candidate)
grid = H.build_grid(H.sweep_axis(0.01, 4.00, 0.01)[:nt], H.sweep_axis(4.000, 5.245, 0.005), (1,), tube_length=n * apix)
for mode in (2, 1, 0):  # fused; run tables + second pass; raster + two transforms per candidate
    eng.set_table_path(mode)
    scores = eng.sweep(grid.params)
    print(eng.last_first_pass, "best", grid.params[int(np.argmax(scores[0]))], "batch", eng.max_batch)
