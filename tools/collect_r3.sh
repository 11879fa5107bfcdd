#!/bin/bash
# Round 3 evidence (run on the GPU box from the repo root): tools/collect_r3.sh <outdir under gpurun_out/>
# The GPU test suite, the bench line with every leg, kernel statistics + SQ counter passes of the general-size sweep at
# 400 x 400 and 200 x 200 (tools/collect_sq_gen.sh), kernel statistics of the bench command, and the general-size fuzz.
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O; export TMPDIR=/tmp
T="timeout -k 10"
$T 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1 || { tail -30 $O/suite.log; exit 1; }
tail -2 $O/suite.log
$T 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
echo bench done
bash tools/collect_sq_gen.sh $1/gen400 400 || exit 1
bash tools/collect_sq_gen.sh $1/gen200 200 || exit 1
cd /tmp
$T 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs > $O/bench_under_rocprof.json 2> $O/stats.log || exit 1
cd $R
$T 600 python tools/fuzz_general.py 40 7 > $O/fuzz_general.txt 2>&1 || { tail -5 $O/fuzz_general.txt; exit 1; }
tail -2 $O/fuzz_general.txt
echo collected
