#!/usr/bin/env bash
# round 3, call k: the market-driven round on the device (auxiliary kernel), full gpu suite, headline check after the Dev struct grew
OUT=gpurun_out/${1:-r03k}; mkdir -p $OUT
export ASCHED_SAFETY_DEADLINE_S=120
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_z_market_round.py tests/test_z_mgpu_words.py tests/test_zzz_market_iterator.py tests/test_z_optimiser.py tests/test_z_optimiser_round.py -q -m gpu -k "market or mgpu or words or optimiser or private or scores or preempts" > $OUT/pytest_market.log 2>&1; echo "pytest(market) rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_market.log | tee -a $OUT/summary.txt
for i in 1 2; do timeout 600 python bench.py --steps 8 --warmup 2 --cpu-budget 0 --no-other > $OUT/bench_headline_$i.json 2> $OUT/bench_headline_$i.err; echo "headline $i rc=$?" | tee -a $OUT/summary.txt; python -c "import json;d=json.load(open('$OUT/bench_headline_$i.json'));print(d['ms_per_step'],d['p50_ms'],d['round']['k_control_ms'])" | tee -a $OUT/summary.txt; done
python tools/round_fingerprint.py 2>&1 | tail -1 | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest(all gpu) rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
