#!/usr/bin/env python3
"""Run-to-run check of the batched Path-A solver in separate processes: prints hashes of the test image, the set-up
(rows, pairs) and the solve (scores, iteration counts) of one group of K candidates."""
import hashlib
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from helicon_amd.solver import PathABatch, hh_pa_params  # noqa: E402
from tools.path_a_bench import L3, NX, NY, TARGET, test_image  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
img = test_image()
h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]  # noqa: E731
tw = np.linspace(27.0, 31.0, k)
params = [hh_pa_params(1.0, float(t), 4.0, 1, 0.0, 0.0, 0.0, NY, NX, NY, 0, L3, TARGET, TARGET, 0, 0, 0) for t in tw]
with PathABatch(img, params) as B:
    rows = h(np.concatenate([B.rhs(c)[1] for c in range(k)]))
    pairs = h(np.concatenate([B.sym_pairs(c) for c in range(k)]))
    x, scores, info = B.solve(np.ones(k, dtype=np.int32), 0)
    print("image", h(img), "rows", rows, "pairs", pairs, "x", h(x), "scores", h(scores), "info", h(info), "iterations", int(info[:, 3].sum()),
          "first", int(info[:, 4].sum()))
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    runs, fails = [(x, scores, info)], [B.counters()["self_check_failures"]]
    for _ in range(reps):
        runs.append(B.solve(np.ones(k, dtype=np.int32), 0))
        fails.append(B.counters()["self_check_failures"])
    print("self-check failures per run:", fails)
    # the majority result per candidate is taken as the reference; report every (run, candidate) that departs from it
    infos = np.stack([r[2] for r in runs])
    xs = np.stack([r[0] for r in runs])
    for c in range(k):
        keys = [xs[r, c].tobytes() for r in range(len(runs))]
        major = max(set(keys), key=keys.count)
        for r in range(len(runs)):
            if keys[r] != major:
                ref = infos[keys.index(major), c]
                print(f"run {r} candidate {c} (c % 8 = {c % 8}): info {infos[r, c].tolist()} vs {ref.tolist()}; max |dx| "
                      f"{np.abs(xs[r, c] - xs[keys.index(major), c]).max():.3e}; dscore {runs[r][1][c] - runs[keys.index(major)][1][c]:.3e}")
    print("runs", len(runs), "distinct x hashes", len({h(r[0]) for r in runs}))
    sys.stdout.flush()
