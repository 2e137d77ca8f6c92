#!/usr/bin/env bash
set -u
OUT=gpurun_out/r04g; mkdir -p $OUT
timeout 600 python tools/probe_fit.py 2>&1 | tail -n 2 | tee -a $OUT/summary.txt
timeout 600 python -m pytest tests -q -m gpu -k "fit or submit or sharded or mgpu" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -n 2 $OUT/pytest.log | tee -a $OUT/summary.txt
