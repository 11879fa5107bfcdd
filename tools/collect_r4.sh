#!/bin/bash
# Round 4 evidence (run on the GPU box from the repo root): tools/collect_r4.sh <A|B|C> <outdir under gpurun_out/>
#   A  the GPU test suite, the bench line with every leg, kernel statistics of the bench command
#   B  counter passes: calibrated FETCH / WRITE traffic and the SQ / LDS counters of the C2 sweep, kernel statistics and
#      traffic of configuration C5 (compact rows of q)
#   C  the trilinear and nearest-neighbour Path A solvers: kernel statistics, SQ counters of one trilinear group
# Counter passes use --pmc alone, statistics passes --kernel-trace --stats alone.  The raw per-dispatch tables are parsed
# here and deleted (only summaries travel back).
set -o pipefail
PART=$1; R=$PWD; O=$R/gpurun_out/$2; mkdir -p $O; export TMPDIR=/tmp
T="timeout -k 10"
stats_csv() { f=$(ls $1/*/*kernel_stats.csv | head -1); cp $f $2; rm -rf $1; }
if [ $PART = A ]; then
  $T 1000 python -m pytest tests -m gpu -q > $O/suite.log 2>&1; tail -3 $O/suite.log
  $T 700 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
  echo bench done
  cd /tmp
  $T 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs > $O/bench_under_rocprof.json 2> $O/stats.log || exit 1
  cd $R; stats_csv $O/stats $O/kernel_stats.csv
  python3 tools/bench_summary.py $O/bench.json > $O/bench_summary.txt 2>&1
  echo collected A
elif [ $PART = B ]; then
  cd /tmp
  $T 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/traffic_run.py > $O/fetch.log 2>&1 || exit 1
  $T 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/traffic_run.py > $O/write.log 2>&1 || exit 1
  cd $R
  python3 tools/traffic_parse.py $O/fetch $O/write "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/traffic_run.py, round 4 ($2)" > $O/traffic_parse.log 2>&1 || exit 1
  cp profiles/traffic.json $O/traffic.json; rm -rf $O/fetch $O/write
  echo traffic done
  bash tools/collect_sq.sh $2/sq512 512 || exit 1
  rm -rf $O/sq512/sq1 $O/sq512/sq2 $O/sq512/sq3
  cd /tmp
  $T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5stats -- python3 $R/tools/c5_prof.py 5 64 > $O/c5_under_rocprof.txt 2>&1 || exit 1
  $T 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c5fetch -- python3 $R/tools/c5_prof.py 2 64 > $O/c5fetch.log 2>&1 || exit 1
  $T 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c5write -- python3 $R/tools/c5_prof.py 2 64 > $O/c5write.log 2>&1 || exit 1
  cd $R; stats_csv $O/c5stats $O/kernel_stats_c5.csv
  python3 tools/c5_traffic.py $O/c5fetch $O/c5write $O/traffic.json > $O/traffic_c5.json 2> $O/traffic_c5.log || exit 1
  rm -rf $O/c5fetch $O/c5write
  $T 120 python3 tools/c5_prof.py 10 64 2>&1 | tail -1 > $O/c5_times.txt
  $T 120 python3 tools/c5_prof.py 10 64 100 2>&1 | tail -1 >> $O/c5_times.txt
  cat $O/c5_times.txt
  echo collected B
else
  cd /tmp
  HH_PAB_TIMING=1 $T 300 python3 $R/tools/path_a_prof.py 256 256 1 linear > $O/lin_timing.log 2>&1
  $T 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/linstats -- python3 $R/tools/path_a_prof.py 256 256 1 linear > $O/lin_under_rocprof.log 2>&1 || exit 1
  $T 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nnstats -- python3 $R/tools/path_a_prof.py 1024 128 8 nn > $O/nn_under_rocprof.log 2>&1 || exit 1
  cd $R; stats_csv $O/linstats $O/path_a_linear_kernel_stats.csv; stats_csv $O/nnstats $O/path_a_kernel_stats.csv
  bash tools/collect_sq_lin.sh $2/sqlin 128 || exit 1
  rm -rf $O/sqlin/sq1 $O/sqlin/sq2 $O/sqlin/sq3
  $T 300 python3 tools/path_a_groups.py 512 --linear > $O/path_a_groups_linear.txt 2>&1
  tail -2 $O/path_a_groups_linear.txt
  $T 300 python3 tools/prep_bench.py 512 > $O/image_prep.json 2> $O/image_prep.err
  tail -2 $O/image_prep.json | cut -c1-400
  echo collected C
fi
