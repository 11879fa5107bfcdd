#!/usr/bin/env python3
"""Debug helper: compare the stopping-test traces (HH_PAB_TRACE / HH_PAB_TRACE_FILE) of repeated solves."""
import glob
import sys

import numpy as np

files = sorted(glob.glob(sys.argv[1] + ".*"), key=lambda f: int(f.rsplit(".", 1)[1]))
tr = [np.fromfile(f).reshape(-1, 8) for f in files]
n = [int(np.nonzero(t.any(axis=1))[0].max()) + 1 for t in tr]
print("rows used per run:", n)
ref = tr[int(np.argmax([n.count(v) for v in n]))]
for k, t in enumerate(tr):
    d = np.nonzero((t != ref).any(axis=1))[0]
    if len(d):
        r = d[0]
        print(f"run {k}: first differing row {r}; cols {np.nonzero(t[r] != ref[r])[0].tolist()}")
        for rr in range(max(0, r - 1), min(r + 3, len(t))):
            print("   this", rr, t[rr].tolist())
            print("   ref ", rr, ref[rr].tolist())
