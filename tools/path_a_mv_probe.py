#!/usr/bin/env python3
"""One candidate's forward product (hh_pab_matvec) of the trilinear batch solver, repeated: wall time per call.  With the
timing-only builds of csrc/path_a_factored.inc (HH_PABF_ABLATE, via HELICON_HIP_LIB) the differences say what a workgroup of
the product spends its time on (one candidate = one round of workgroups, so a call's device time is a workgroup's)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from helicon_amd.solver import PathABatch, hh_pa_params  # noqa: E402
from tools.path_a_bench import NX, NY, L3, TARGET, test_image  # noqa: E402

if __name__ == "__main__":
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    img = test_image()
    params = [hh_pa_params(1.0, float(t), 4.0, 1, 0.0, 0.0, 0.0, NY, NX, NY, 0, L3, TARGET, TARGET, 1, 0, 0) for t in np.linspace(28.0, 30.0, k)]
    B = PathABatch(img, params)
    x = np.random.default_rng(0).random(B.n)
    B.matvec(0, x)
    t0 = time.perf_counter()
    for r in range(reps):
        y = B.matvec(r % k, x)
    dt = (time.perf_counter() - t0) / reps
    print(f"forward product of one candidate: {dt * 1e6:.1f} us per call (wall, {reps} calls), |y| = {float(np.linalg.norm(y)):.6g}")
