#!/usr/bin/env bash
# round 4, second call: the communicator on the MI355X (RCCL world size 1, external transport with device buffers), the multi-process tests, and the at-scale
# differential rounds on -O2 / -O1 / -Os builds of the round kernel ("at-scale parity tests green under -O1 / -O2 / -Os builds")
set -u
OUT=gpurun_out/r04b; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -k "comm_in_library or lds_idiom or sharded_fit or mgpu_words or queuehash" -p no:cacheprovider > $OUT/pytest_comm.log 2>&1; echo "comm pytest rc=$?" | tee -a $OUT/summary.txt; tail -n 6 $OUT/pytest_comm.log | tee -a $OUT/summary.txt
for V in O2 O1 Os; do
  ASCHED_LIB_PATH=$PWD/armada_amd/csrc/libarmada_sched_$V.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "at_scale or 64k or random_rounds_match" -p no:cacheprovider > $OUT/pytest_$V.log 2>&1
  echo "-$V at-scale pytest rc=$?" | tee -a $OUT/summary.txt; tail -n 3 $OUT/pytest_$V.log | tee -a $OUT/summary.txt
done
