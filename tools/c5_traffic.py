#!/usr/bin/env python3
"""HBM traffic of configuration C5's two kernels from the FETCH_SIZE / WRITE_SIZE passes over tools/c5_prof.py, converted
with the calibration of the C2 traffic passes (profiles/traffic.json: bytes per counter unit for this device):
    python3 tools/c5_traffic.py <fetch dir> <write dir> <traffic.json>  ->  JSON on stdout (per launch, 20,000 candidates)."""
import json
import sys

from traffic_parse import per_kernel

if __name__ == "__main__":
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    cal = json.load(open(sys.argv[3]))["calibration"]
    fu, wu = cal["bytes_per_FETCH_SIZE_unit"], cal["bytes_per_WRITE_SIZE_unit"]
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/c5_prof.py (64 segments x 20,000 candidates), calibration of profiles/traffic.json",
           "candidates_per_launch": 20000, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if "k_fused_pass" not in k and "k_segment_corr" not in k:
            continue
        f, w = fetch.get(k, (0.0, 0))[0] * fu, write.get(k, (0.0, 0))[0] * wu
        out["kernels"][k[:100]] = {"fetched_bytes_per_launch": f, "written_bytes_per_launch": w, "launches_seen": fetch.get(k, (0, 0))[1],
                                   "fetched_bytes_per_candidate": f / 20000, "written_bytes_per_candidate": w / 20000}
    print(json.dumps(out, indent=1))
