#!/usr/bin/env python3
"""Short table of a bench.py JSON line: headline, roofline fraction, and every leg's rate with its checks."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"]), d["unit"], "| ms/step", round(d["ms_per_step"], 3), "| frac", round(d.get("roofline", {}).get("frac", 0), 4),
      "| argmax", d.get("argmax", {}).get("is_truth"), "| cpu", d.get("cpu_baseline", {}).get("value"))
keys = ("value", "error", "ms_per_step", "segments_at_truth", "segments", "argmax_equals_host_argmax", "oracle_max_abs_err", "check_failed",
        "truth_rank_segment0", "best_twist_rise_segment0", "best_twist", "oracle_abs_err", "best_is_truth", "seconds", "oracle_seconds_for_one_candidate", "self_check_failures")
for k, v in d.get("pipelines", {}).items():
    if isinstance(v, dict):
        row = {kk: (float(f'{vv:.3g}') if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in keys}
        if "roofline" in v and isinstance(v["roofline"], dict):
            row["frac"] = round(v["roofline"].get("frac", 0), 3)
        print(k, row)
