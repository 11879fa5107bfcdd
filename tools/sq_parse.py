#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc SQ_* passes over tools/traffic_run.py per kernel:
    python tools/sq_parse.py <dir> [<dir> ...] > profiles/rNN_pmc_sq.json
Averages are per launch.  SQ_INSTS_* count wave-instructions; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count
quad-cycles (MI355X_MICROARCH.md, cycle-constant table); SQ_LDS_* count LDS-array cycles."""
import csv
import glob
import json
import sys
from collections import defaultdict

KERNELS = ("k_fused_pass<512", "k_second_pass<512, 0, 1>", "k_first_pass_table<512>", "k_first_pass<512, 0>",
           "k_run_table<512>", "k_column_factors<512>", "k_fused_pass<1024", "k_second_pass<1024, 0, 1>",
           "k_first_pass_table<1024>", "k_first_pass<1024, 0>", "k_gen_rows<", "k_gen_fused<",
           "k_segment_corr", "k_pabs_matvec<1", "k_pabs_rmatvec<1", "k_pabs_matvec<0", "k_pabs_matvec<2", "k_pabs_rmatvec<0",
           "k_pabf_matvec<1", "k_pabf_scatter<1", "k_pabl_finish<1", "k_pabf_tail<1", "k_pabl_matvec<1", "k_pabl_scatter<1")
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
dur = defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            k = next((s for s in KERNELS if s in row["Kernel_Name"]), None)
            if k is None:
                continue
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
            key = (f, row["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
out = {}
for k in tot:
    e = {c: tot[k][c] / cnt[k][c] for c in tot[k]}
    e["launches"] = max(cnt[k].values())
    e["avg_us_under_pmc"] = sum(dur[k]) / len(dur[k]) / 1e3
    if "SQ_INSTS_VALU" in e and "SQ_ACTIVE_INST_VALU" in e:
        # one SIMD issues one wave64 vector instruction per quad-cycle at best: busy fraction of the 1024 SIMDs
        cycles = e["avg_us_under_pmc"] * 1e-6 * 2.4e9
        e["valu_quadcycles_per_simd_over_kernel_cycles_at_2.4GHz"] = e["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cycles
    if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_share"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
    out[k] = e
print(json.dumps(out, indent=1))
