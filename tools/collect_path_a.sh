#!/bin/bash
# Path A evidence of round 3 (run on the GPU box from the repo root): tools/collect_path_a.sh <outdir under gpurun_out/>
# Kernel stats and the two traffic-counter passes over one group of 256 candidates, then the throughput table.
# Counter passes use --pmc alone; the stats pass --kernel-trace --stats alone.
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O; export TMPDIR=/tmp
T="timeout -k 10"
cd /tmp
$T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/path_a_traffic.py 256 > $O/run_stats.json 2> $O/stats.log || exit 1
echo stats done >> $O/progress.log
$T 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/path_a_traffic.py 256 > $O/run_fetch.json 2> $O/fetch.log || exit 1
$T 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/path_a_traffic.py 256 > $O/run_write.json 2> $O/write.log || exit 1
echo traffic done >> $O/progress.log
cd $R
python3 tools/path_a_traffic.py parse $O/fetch $O/write $O/stats $O/run_stats.json > $O/parse.log 2>&1 || exit 1
cp profiles/r03_path_a_traffic.json $O/
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
$T 600 python3 tools/path_a_bench.py > $O/path_a_bench.txt 2>&1 || exit 1
echo collected
