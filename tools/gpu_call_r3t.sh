#!/usr/bin/env bash
# round 3, call t: does the code of fastStreamPrepareOne (flag off) move the headline?  previous commit's library vs this one, interleaved
OUT=gpurun_out/${1:-r03t}; mkdir -p $OUT
for rep in 1 2 3; do
  for lib in armada_amd/csrc/libarmada_sched_prev.so armada_amd/csrc/libarmada_sched.so; do
    name=$(basename $lib .so); name=${name#libarmada_sched}; name=${name:-_current}
    ASCHED_LIB_PATH=$PWD/$lib timeout 300 python bench.py --steps 8 --warmup 2 --cpu-budget 0 --no-other > $OUT/b.json 2> $OUT/b.err
    python -c "import json;d=json.load(open('$OUT/b.json'));print('$name rep $rep', round(d['ms_per_step'],1), round(d['p50_ms'],1), 'k_control', round(d['round']['k_control_ms'],1))" | tee -a $OUT/summary.txt
  done
done
