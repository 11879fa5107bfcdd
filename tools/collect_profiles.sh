#!/bin/bash
# Collect the evidence kept under profiles/ (run on the GPU box from the repo root):
#   tools/collect_profiles.sh <outdir under gpurun_out/>
# The default bench line (with its run-tables / transform / C3 legs and the CPU baseline), rocprofv3 kernel stats of the
# same bench command (fused pipeline, N = 512) and of the C4 configuration (N = 1024), the two calibrated traffic
# passes (FETCH_SIZE, WRITE_SIZE) and the SQ counter passes over tools/traffic_run.py at 512 and 1024, and the other
# BASELINE configurations.  Counter passes use --pmc alone; stats passes use --kernel-trace --stats alone.
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O; export TMPDIR=/tmp
T="timeout -k 10"
$T 500 python bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo bench done >> $O/progress.log
cd /tmp
$T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --no-extra-legs > $O/bench_under_rocprof.json 2> $O/stats.log || exit 1
echo stats512 done >> $O/progress.log
$T 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1024 -- python $R/tools/configs.py C4 > $O/c4_under_rocprof.txt 2> $O/stats1024.log || exit 1
echo stats1024 done >> $O/progress.log
$T 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/tools/traffic_run.py > $O/fetch.log 2>&1 || exit 1
$T 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/tools/traffic_run.py > $O/write.log 2>&1 || exit 1
echo traffic done >> $O/progress.log
cd $R
bash tools/collect_sq.sh $1/sq512 512 || exit 1
echo sq512 done >> $O/progress.log
bash tools/collect_sq.sh $1/sq1024 1024 || exit 1
echo sq1024 done >> $O/progress.log
$T 600 python tools/configs.py > $O/configs.txt 2>&1 || exit 1
python tools/traffic_parse.py $O/fetch $O/write "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/traffic_run.py, round 2 ($1)" > $O/traffic_parse.log 2>&1 || exit 1
cp profiles/traffic.json $O/traffic.json
echo collected
