#!/bin/bash
# SQ / LDS counter passes over tools/traffic_run.py (run on the GPU box from the repo root):
#   tools/collect_sq.sh <outdir under gpurun_out/> [image side]
# Counter passes only (--pmc with nothing else), one set per rocprofv3 run; aggregate with tools/sq_parse.py.
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; N=${2:-512}; mkdir -p $O; export TMPDIR=/tmp
T="timeout -k 10"
cd /tmp
$T 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- python $R/tools/traffic_run.py $N > $O/sq1.log 2>&1 || exit 1
$T 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python $R/tools/traffic_run.py $N > $O/sq2.log 2>&1 || exit 1
$T 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $O/sq3 -- python $R/tools/traffic_run.py $N > $O/sq3.log 2>&1 || exit 1
cd $R
python tools/sq_parse.py $O/sq1 $O/sq2 $O/sq3 > $O/pmc_sq.json
echo collected
