#!/bin/bash
# SQ / LDS counter passes over one group of trilinear Path A candidates (tools/path_a_prof.py K K 1 linear), on the GPU box:
#   tools/collect_sq_lin.sh <outdir under gpurun_out/> [K]
# Counter passes use --pmc alone.  Aggregate: tools/sq_parse.py.
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; K=${2:-128}; mkdir -p $O; export TMPDIR=/tmp
T="timeout -k 10"
cd /tmp
$T 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- python3 $R/tools/path_a_prof.py $K $K 1 linear > $O/sq1.log 2>&1 || exit 1
$T 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN --output-format csv -d $O/sq2 -- python3 $R/tools/path_a_prof.py $K $K 1 linear > $O/sq2.log 2>&1 || exit 1
$T 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $O/sq3 -- python3 $R/tools/path_a_prof.py $K $K 1 linear > $O/sq3.log 2>&1 || exit 1
cd $R
python3 tools/sq_parse.py $O/sq1 $O/sq2 $O/sq3 > $O/pmc_sq.json
echo collected
