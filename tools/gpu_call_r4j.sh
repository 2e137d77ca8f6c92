#!/usr/bin/env bash
# the wide passes with grouped loads (scanPart / fairPart): parity at scale, then A/B against the previous commit's library ("prev")
set -u
OUT=gpurun_out/r04j; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -k "at_scale or 64k or preempt or nodedb or goldens" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -n 3 $OUT/pytest.log | tee -a $OUT/summary.txt
for L in prev new; do
  P=$PWD/armada_amd/csrc/libarmada_sched_$L.so; [ $L = new ] && P=$PWD/armada_amd/csrc/libarmada_sched.so
  for W in "" full; do
    echo "== $L config4 $W" | tee -a $OUT/summary.txt
    ASCHED_LIB_PATH=$P timeout 400 python tools/prof_config4.py $W 2>&1 | grep "^round" | tail -n 1 | cut -c1-330 | tee -a $OUT/summary.txt
  done
done
