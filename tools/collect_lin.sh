cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
HH_PAB_TIMING=1 timeout -k 10 300 python3 $R/tools/path_a_prof.py 256 256 1 linear > $R/gpurun_out/r4_lin_timing.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_linprof1 -- python3 $R/tools/path_a_prof.py 256 256 1 linear > $R/gpurun_out/r4_linprof1.log 2>&1
cd $R
f=$(ls gpurun_out/r4_linprof1/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r4_linprof1_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/r4_linprof1_kernel_stats.csv")))
for r in rows[:12]:
    print(r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
grep -v "^hh_pab_create\[16\]" gpurun_out/r4_lin_timing.log | tail -12
