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
