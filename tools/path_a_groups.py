#!/usr/bin/env python3
"""lsq_reconstruct_batch over 1,024 (or argv[1]) candidates of the 64 x 128 bench image for several (group size, streams)
settings: wall time, launch counters, and a hash of the scores — every setting must give the same scores, bit for bit.
`--linear` uses the trilinear projector (the app's default)."""
import hashlib
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from helicon_amd.solver import lsq_reconstruct, lsq_reconstruct_batch  # noqa: E402
from tools.path_a_bench import KW, test_image  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
total = int(args[0]) if args else 1024
interp = "linear" if "--linear" in sys.argv else "nn"
image = test_image()
lsq_reconstruct(image, 1.0, 29.0, 4.0, 1, interpolation=interp, **KW)   # warm
tw = np.linspace(27.0, 31.0, total)
cands = [(float(t), 4.0, 1) for t in tw]
settings = [(total, 1), (total // 2, 2), (total // 4, 4), (128, 8)] if "--all" in sys.argv else [(total, 1), (total // 2, 2)]
for batch, streams in settings:
    st = {}
    t0 = time.perf_counter()
    res = lsq_reconstruct_batch(image, 1.0, cands, return_3d=False, batch=batch, streams=streams, stats=st, interpolation=interp, **KW)
    dt = time.perf_counter() - t0
    scores = np.array([s for _, s in res])
    info = np.array(st["info"])
    print(f"{interp}: {total} candidates, groups of {batch} on {streams} stream(s): {dt:.3f} s = {total / dt:.1f} candidates/s; "
          f"{st['launches']} launches, {st['host_syncs']} host syncs, LSMR iterations {int(info[:, 3].sum())} (max {int(info[:, 3].max())}); "
          f"self-check {st['self_check_failures']}; scores {hashlib.sha1(scores.tobytes()).hexdigest()[:12]}; best {tw[int(np.argmax(scores))]:.3f} "
          f"({scores.max():.4f})", flush=True)
