#!/usr/bin/env bash
# round 3, call j: the multi-GPU exchange words on the device (N = 1: buffers on the GPU, collective skipped), new gpu tests, headline check
OUT=gpurun_out/${1:-r03j}; mkdir -p $OUT
export ASCHED_SAFETY_DEADLINE_S=120
timeout 900 python -m pytest tests/test_z_mgpu_words.py tests/test_sharded_fit.py tests/test_queuehash.py tests/test_abi.py -q -m gpu -x > $OUT/pytest_new.log 2>&1; echo "pytest(new) rc=$?" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-budget 0 --no-other > $OUT/bench_headline.json 2> $OUT/bench_headline.err; echo "headline rc=$?" | tee -a $OUT/summary.txt
timeout 600 python bench.py --mode node-sharded-fit --steps 20 --warmup 3 > $OUT/bench_mode_node_sharded_fit.json 2> $OUT/bench_mode_nsf.err; echo "nsf rc=$?" | tee -a $OUT/summary.txt
timeout 600 python bench.py --mode queue-hash --steps 3 --warmup 1 --nodes 20000 --jobs 200000 --queues 32 > $OUT/bench_mode_queue_hash_20k.json 2> $OUT/bench_mode_qh20.err; echo "qh20k rc=$?" | tee -a $OUT/summary.txt
timeout 900 python bench.py --mode queue-hash --steps 2 --warmup 1 > $OUT/bench_mode_queue_hash_full.json 2> $OUT/bench_mode_qhfull.err; echo "qhfull rc=$?" | tee -a $OUT/summary.txt
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_modes" -- python "$OLDPWD/bench.py" --mode queue-hash --steps 2 --warmup 1 --nodes 20000 --jobs 200000 --queues 32 > "$OLDPWD/$OUT/prof_modes.log" 2>&1 )
DB=$(find "$OUT/prof_modes" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py "$DB" "$OUT/kernel_stats_mode_queue_hash_20k.csv" > /dev/null
find "$OUT" -name "*.db" -size +8M -delete
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest(all gpu) rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
for f in $OUT/bench_headline.json $OUT/bench_mode_*.json; do echo "== $f"; head -c 1500 $f; echo; done | tee -a $OUT/summary.txt
