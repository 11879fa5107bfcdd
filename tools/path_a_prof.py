#!/usr/bin/env python3
"""One batch of K Path-A candidates (tools/path_a_bench.py's 64 x 128 case) for rocprofv3: `--kernel-trace --stats`."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tools.path_a_bench import batch_run, test_image  # noqa: E402

if __name__ == "__main__":
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    r = batch_run(test_image(), k, repeat=1)
    print({key: r[key] for key in ("k", "setup_s", "solve_s", "lsmr_iterations", "launches", "host_syncs", "lsmr_iterations_queued")})
