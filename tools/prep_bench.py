#!/usr/bin/env python3
"""Measurement of the on-device image preparation: Gaussian low/high pass and
threshold for one N x N image, host buffers in and out (the boundary's form), with the NumPy oracle timed
beside it.  Prints one JSON line per size."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import helicon_amd as H  # noqa: E402
from oracle import path_b as O  # noqa: E402  (CPU baseline only)

for n in [int(a) for a in sys.argv[1:]] or [512, 1024]:
    img = (np.random.default_rng(n).normal(size=(n, n)) + 2.0).astype(np.float32)
    eng = H.SweepEngine(n)
    eng.low_high_pass_filter(img, 0.25, 2.0 / n)  # warm-up
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        got = eng.low_high_pass_filter(img, 0.25, 2.0 / n)
    gpu_ms = 1e3 * (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(5):
        ref = O.low_high_pass_filter(img.astype(np.float64), 0.25, 2.0 / n)
    cpu_ms = 1e3 * (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(reps):
        thr = eng.threshold_data(got, thresh_fraction=0.1)
    thr_ms = 1e3 * (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(5):
        thr_ref = O.threshold_data(got, thresh_fraction=np.float32(0.1))
    thr_cpu_ms = 1e3 * (time.perf_counter() - t0) / 5
    print(json.dumps({
        "op": "low_high_pass_filter + threshold_data", "n": n,
        "filter_ms_host_to_host": gpu_ms, "filter_ms_numpy_oracle_1core": cpu_ms,
        "filter_max_abs_err": float(np.abs(got - ref).max()), "image_max_abs": float(np.abs(img).max()),
        "threshold_ms_host_to_host": thr_ms, "threshold_ms_numpy": thr_cpu_ms,
        "threshold_exact": bool(np.array_equal(thr, thr_ref)),
        "note": "host-to-host times include 2 x N^2 x 4 B over PCIe and a stream synchronise; the transforms themselves "
                "are four launches of N/2..N/2+1 single-transform workgroups",
    }), flush=True)
    eng.close()

# ---- round 4: the scikit-image part of the preparation (csrc/image_prep.inc) against oracle/prep.py on the host ------------------
from helicon_amd import denovo3D as D  # noqa: E402
from oracle import prep as P  # noqa: E402  (CPU baseline and checker only)


def _timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return 1e3 * (time.perf_counter() - t0) / reps, out


for ny, nx in [(256, 512), (512, 1024)]:
    yy, xx = np.mgrid[0:ny, 0:nx].astype(np.float64)
    a = np.deg2rad(6.0)
    dist = -(xx - nx / 2) * np.sin(a) + (yy - ny / 2 - 4.0) * np.cos(a)
    img = (np.exp(-0.5 * (dist / (ny / 12)) ** 2) * (np.abs(dist) < ny / 5) * (1 + 0.05 * np.random.default_rng(1).random((ny, nx)))).astype(np.float32)
    row = {"op": "scikit-image part of the image preparation", "shape": [ny, nx]}
    for name, dev, cpu in [
        ("down_scale_x0.4", lambda: D.down_scale(img, 2.5, 1.0), lambda: P.down_scale(img, 2.5, 1.0)),
        ("transform_image_rot7", lambda: D.transform_image(img, rotation=7.0, post_translation=(3.0, 0.0)),
         lambda: P.transform_image(img, rotation=7.0, post_translation=(3.0, 0.0))),
        ("estimate_helix", lambda: D.estimate_helix_rotation_center_diameter(img), lambda: P.estimate_helix_rotation_center_diameter(img)),
        ("rotate_shift_cubic", lambda: D.rotate_shift_image(img, 7.0, (0, 0), (3.0, 0), order=3), lambda: P.rotate_shift_image(img, 7.0, (0, 0), (3.0, 0), order=3)),
    ]:
        g_ms, g = _timed(dev, 20)
        c_ms, c = _timed(cpu, 3)
        err = float(np.abs(np.asarray(g, dtype=np.float64) - np.asarray(c, dtype=np.float64)).max())
        row[name] = {"device_ms_host_to_host": round(g_ms, 3), "scipy_numpy_ms_1core": round(c_ms, 3), "max_abs_diff": err}
    print(json.dumps(row), flush=True)
