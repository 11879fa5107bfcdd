#!/usr/bin/env python3
"""Path A evidence for profiles/: workload and parser.

  run   (default)  calibration launches (1 GiB read / written with known shapes, like tools/traffic_run.py), then ONE group
                   of K candidates of tools/path_a_bench.py's 64 x 128 case through hh_pab_create + hh_pab_solve, its
                   counters printed as JSON.  Under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (one pass each) or
                   `--kernel-trace --stats`.
  parse FETCH_DIR WRITE_DIR STATS_DIR RUN_JSON
                   -> profiles/r03_path_a_traffic.json: bytes the two LSMR products really move per candidate-iteration
                   (counters converted with the calibration of the same run), next to the algorithmic figure, and the
                   kernels' durations from the stats pass."""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
CALIB_BYTES = 1 << 30


def run(k):
    import numpy as np

    import helicon_amd as H
    from tools.path_a_bench import batch_run, test_image

    eng = H.SweepEngine(512)
    for _ in range(3):
        eng.calibrate_traffic(0, CALIB_BYTES)
        eng.calibrate_traffic(1, CALIB_BYTES)
    r = batch_run(test_image(), k, repeat=1)
    print(json.dumps(r))


def counters(dirname, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    tot[row["Kernel_Name"]] += float(row["Counter_Value"])
                    cnt[row["Kernel_Name"]] += 1
    return tot, cnt


def pick(d, frag):
    return sum(v for k, v in d.items() if frag in k)


def parse(fetch_dir, write_dir, stats_dir, run_json):
    run_info = json.loads(Path(run_json).read_text().strip().splitlines()[-1])
    ft, fc = counters(fetch_dir, "FETCH_SIZE")
    wt, wc = counters(write_dir, "WRITE_SIZE")
    f_unit = CALIB_BYTES / (pick(ft, "k_calib_read") / pick(fc, "k_calib_read"))
    w_unit = CALIB_BYTES / (pick(wt, "k_calib_write") / pick(wc, "k_calib_write"))
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --kernel-trace --stats over tools/path_a_traffic.py "
                     f"(one group of {run_info['k']} candidates, one stream), round 3",
           "calibration": {"bytes_per_FETCH_SIZE_unit": f_unit, "bytes_per_WRITE_SIZE_unit": w_unit},
           "run": run_info}
    iters = run_info["lsmr_iterations"]
    kernels = {}
    for name, frag in (("lsmr_forward", "k_pabs_matvec<1"), ("lsmr_transposed", "k_pabs_rmatvec<1")):
        rb, wb = pick(ft, frag) * f_unit, pick(wt, frag) * w_unit
        kernels[name] = {"launches": int(pick(fc, frag)), "read_bytes": rb, "write_bytes": wb,
                         "bytes_per_candidate_iteration": (rb + wb) / iters}
    for f in glob.glob(f"{stats_dir}/**/*kernel_stats.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                for name, frag in (("lsmr_forward", "k_pabs_matvec<1"), ("lsmr_transposed", "k_pabs_rmatvec<1")):
                    if frag in row["Name"]:
                        k = kernels[name]
                        k["stats_calls"] = k.get("stats_calls", 0) + int(row["Calls"])
                        k["stats_total_ns"] = k.get("stats_total_ns", 0) + int(row["TotalDurationNs"])
    for k in kernels.values():
        if "stats_total_ns" in k:
            k["avg_us"] = k["stats_total_ns"] / k["stats_calls"] / 1e3
            k["us_per_candidate_iteration"] = k["stats_total_ns"] / 1e3 / iters
            k["measured_TBps"] = (k["read_bytes"] + k["write_bytes"]) / k["stats_total_ns"] / 1e3
    out["kernels"] = kernels
    out["measured_bytes_per_candidate_iteration"] = sum(k["bytes_per_candidate_iteration"] for k in kernels.values())
    out["algorithmic_bytes_per_candidate_iteration"] = run_info["bytes_per_lsmr_iteration"]
    (ROOT / "profiles" / "r03_path_a_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "parse":
        parse(*sys.argv[2:6])
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
