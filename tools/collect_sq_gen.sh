#!/bin/bash
# SQ / LDS counter passes and the kernel statistics of the general-size sweep (tools/general_prof.py), on the GPU box:
#   tools/collect_sq_gen.sh <outdir under gpurun_out/> <image side>
# Counter passes use --pmc alone; the stats pass --kernel-trace --stats alone.  Aggregate: tools/sq_parse.py.
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; N=${2:-400}; mkdir -p $O; export TMPDIR=/tmp
T="timeout -k 10"
cd /tmp
$T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/general_prof.py $N 5 > $O/stats.log 2>&1 || exit 1
$T 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- python3 $R/tools/general_prof.py $N 2 > $O/sq1.log 2>&1 || exit 1
$T 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 $R/tools/general_prof.py $N 2 > $O/sq2.log 2>&1 || exit 1
$T 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $O/sq3 -- python3 $R/tools/general_prof.py $N 2 > $O/sq3.log 2>&1 || exit 1
$T 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INSTS_MISC --output-format csv -d $O/sq4 -- python3 $R/tools/general_prof.py $N 2 > $O/sq4.log 2>&1 || exit 1
cd $R
python3 tools/sq_parse.py $O/sq1 $O/sq2 $O/sq3 $O/sq4 > $O/pmc_sq.json
grep candidates $O/stats.log
echo collected
