#!/usr/bin/env bash
# round 3, call a: minsize bisect + full gpu suite + bench line
set -u
OUT=gpurun_out/r3a; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
bash tools/minsize_bisect.sh run > $OUT/bisect_stdout.txt 2>&1; cp gpurun_out/minsize/bisect.txt $OUT/ 2>/dev/null
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=12 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -22 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
head -c 1200 $OUT/bench_full.json | tee -a $OUT/summary.txt
cat $OUT/bisect.txt
