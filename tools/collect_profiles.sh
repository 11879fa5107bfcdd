#!/bin/bash
# Collect the evidence kept under profiles/ (run on the GPU box from the repo root):
#   tools/collect_profiles.sh <outdir under gpurun_out/>
# bench lines for the three pipelines, rocprofv3 kernel stats of the default bench, the two
# calibrated traffic passes (FETCH_SIZE, WRITE_SIZE) and four passes of SQ counters over
# tools/traffic_run.py, and the other BASELINE configurations.
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O; export TMPDIR=/tmp
T="timeout -k 10"
$T 400 python bench.py > $O/bench.json 2> $O/bench.err || exit 1
$T 200 python bench.py --no-cpu-baseline --first-pass tables > $O/bench_tables.json 2>> $O/bench.err || exit 1
$T 200 python bench.py --no-cpu-baseline --first-pass transform > $O/bench_transform.json 2>> $O/bench.err || exit 1
cd /tmp
$T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log || exit 1
$T 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/tools/traffic_run.py > $O/fetch.log 2>&1 || exit 1
$T 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/tools/traffic_run.py > $O/write.log 2>&1 || exit 1
$T 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- python $R/tools/traffic_run.py > $O/sq1.log 2>&1 || exit 1
$T 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python $R/tools/traffic_run.py > $O/sq2.log 2>&1 || exit 1
$T 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $O/sq3 -- python $R/tools/traffic_run.py > $O/sq3.log 2>&1 || exit 1
$T 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/sq4 -- python $R/tools/traffic_run.py > $O/sq4.log 2>&1 || exit 1
cd $R
$T 600 python tools/configs.py > $O/configs.txt 2>&1 || exit 1
echo collected
