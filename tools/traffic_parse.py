#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of tools/traffic_run.py into profiles/traffic.json.

FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of the L2's fabric-side request counters
(MI355X_MICROARCH.md, HBM).  The calibration launches move a known byte count with the sweep's own
access shapes, which gives bytes-per-counter-unit for exactly those shapes; the sweep kernels'
counters are then converted with that factor."""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

CALIB_BYTES = 1 << 30
SWEEP_CANDIDATES = 100000
ROOT = Path(__file__).resolve().parent.parent


def per_kernel(dirname, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                k = row["Kernel_Name"]
                tot[k] += float(row["Counter_Value"])
                cnt[k] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def pick(d, frag):
    for k, v in d.items():
        if frag in k:
            return v
    raise KeyError(frag)


def main(fetch_dir, write_dir, source=None, n=512, batch=256, out=None):
    fetch = per_kernel(fetch_dir, "FETCH_SIZE")
    write = per_kernel(write_dir, "WRITE_SIZE")
    f_unit = CALIB_BYTES / pick(fetch, "k_calib_read")[0]    # bytes per FETCH_SIZE unit, K_B's read shape
    w_unit = CALIB_BYTES / pick(write, "k_calib_write")[0]   # bytes per WRITE_SIZE unit, K_A's write shape
    res = {"source": source or "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/traffic_run.py",
           "calibration": {"bytes": CALIB_BYTES, "bytes_per_FETCH_SIZE_unit": f_unit,
                           "bytes_per_WRITE_SIZE_unit": w_unit,
                           "calib_read_FETCH_SIZE": pick(fetch, "k_calib_read")[0],
                           "calib_write_WRITE_SIZE": pick(write, "k_calib_write")[0],
                           "calib_read_WRITE_SIZE": pick(write, "k_calib_read")[0],
                           "calib_write_FETCH_SIZE": pick(fetch, "k_calib_write")[0]}}
    kernels = {}
    for name, frag in (("first_pass", f"k_first_pass<{n}, 0>"), ("second_pass", f"k_second_pass<{n}, 0, 1>"),
                       ("first_pass_table", f"k_first_pass_table<{n}>"), ("fused_pass", f"k_fused_pass<{n}, 0, 1>"),
                       ("column_factors", f"k_column_factors<{n}>"), ("run_table", f"k_run_table<{n}>")):
        try:
            fv, fl = pick(fetch, frag)
            wv, wl = pick(write, frag)
        except KeyError:
            continue
        # tools/traffic_run.py sweeps the 100k-candidate C2 grid once per pipeline; k_second_pass serves two of them
        cands = SWEEP_CANDIDATES * (2 if name == "second_pass" else 1)
        total = (fv * f_unit + wv * w_unit) * fl
        kernels[name] = {"launches": fl, "FETCH_SIZE": fv, "WRITE_SIZE": wv,
                         "read_bytes_per_launch": fv * f_unit, "write_bytes_per_launch": wv * w_unit,
                         "bytes_per_launch": fv * f_unit + wv * w_unit, "candidates": cands,
                         "bytes_per_candidate": total / cands}
    # launch sizes differ between pipelines (and the last launch of a sweep is short), so the figure bench.py
    # uses is per candidate
    res[f"n{n}"] = {k: v["bytes_per_candidate"] for k, v in kernels.items()}
    res[f"n{n}_detail"] = kernels
    out = Path(out) if out else ROOT / "profiles" / "traffic.json"
    out.write_text(json.dumps(res, indent=1) + "\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
