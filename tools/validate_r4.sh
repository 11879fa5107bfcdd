#!/bin/bash
# Long validation campaign of round 4's final code (run on the GPU box from the repo root): tools/validate_r4.sh <outdir under gpurun_out/>
# Bigger fuzz runs than the suite's, and run-to-run reproducibility of both batched Path-A solvers in separate processes.
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
T="timeout -k 10"
$T 600 python tools/fuzz_prep.py 400 21 2>&1 | grep -v amdgpu.ids > $O/fuzz_prep.txt; tail -1 $O/fuzz_prep.txt | cut -c1-300
$T 900 python tools/fuzz_general.py 160 23 2>&1 | grep -v amdgpu.ids | tail -2 > $O/fuzz_general.txt; tail -1 $O/fuzz_general.txt
$T 900 python tools/fuzz_pipelines.py 2>&1 | grep -v amdgpu.ids | tail -2 > $O/fuzz_pipelines.txt; tail -1 $O/fuzz_pipelines.txt
for i in 1 2 3; do $T 300 python tools/path_a_repro.py 256 2 2>&1 | grep -v amdgpu.ids | tail -2; done > $O/repro_nn_256.txt; sort -u $O/repro_nn_256.txt | head -4
for i in 1 2 3; do $T 300 python tools/path_a_groups.py 256 --linear 2>&1 | grep -v "amdgpu.ids\|hh_pab_create" | tail -2; done > $O/repro_linear_256.txt; sed 's/[0-9.]* s = [0-9.]* candidates.s/-/' $O/repro_linear_256.txt | sort -u | cut -c1-200 | head -4
echo validated
