cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in compact full; do
  if [ $mode = full ]; then export HH_Q_FULL=1; else unset HH_Q_FULL; fi
  python3 $R/tools/c5_prof.py 10 64 2>&1 | tail -1
  python3 $R/tools/c5_prof.py 10 64 100 2>&1 | tail -1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_c5prof_$mode -- python3 $R/tools/c5_prof.py 5 64 > /dev/null 2>&1
  f=$(ls $R/gpurun_out/r4_c5prof_$mode/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/r4_c5_${mode}_kernel_stats.csv
  python3 - <<PY
import csv
for r in list(csv.DictReader(open("$R/gpurun_out/r4_c5_${mode}_kernel_stats.csv")))[:4]:
    print("$mode", r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
done
