#!/usr/bin/env bash
# round 4: the whole GPU suite + the full default bench line on the working tree
set -u
OUT=gpurun_out/r04e; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -n 16 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
( time timeout 1500 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err ) 2>> $OUT/summary.txt; echo "bench rc=$?" | tee -a $OUT/summary.txt
head -c 1200 $OUT/bench_full.json | tee -a $OUT/summary.txt
