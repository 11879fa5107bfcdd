#!/usr/bin/env python3
"""lsq_reconstruct_batch over N candidates (tools/path_a_bench.py's 64 x 128 case) for rocprofv3 `--kernel-trace --stats`:
argv = [candidates, group size, streams, nn | linear]."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from helicon_amd.solver import lsq_reconstruct_batch  # noqa: E402
from tools.path_a_bench import KW, test_image  # noqa: E402

if __name__ == "__main__":
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    streams = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    interp = sys.argv[4] if len(sys.argv) > 4 else "nn"
    img = test_image()
    tw = np.linspace(27.0, 31.0, total)
    lsq_reconstruct_batch(img, 1.0, [(float(t), 4.0, 1) for t in tw[:: max(1, total // 16)]], return_3d=False, interpolation=interp, **KW)
    t0 = time.perf_counter()
    lsq_reconstruct_batch(img, 1.0, [(float(t), 4.0, 1) for t in tw], return_3d=False, batch=batch, streams=streams, interpolation=interp, **KW)
    print("seconds", time.perf_counter() - t0)
