#!/usr/bin/env bash
set -u
OUT=gpurun_out/r04i; mkdir -p $OUT
echo "== prof config4 reduced" | tee -a $OUT/summary.txt
ASCHED_PRINT_SEG=1 ASCHED_LIB_PATH=$PWD/armada_amd/csrc/libarmada_sched_prof.so timeout 400 python tools/prof_config4.py 2>&1 | tail -n 4 | cut -c1-900 | tee -a $OUT/summary.txt
