#!/usr/bin/env bash
# on the GPU box: the headline round of every libarmada_sched_pad_*.so and of the default build, interleaved twice (boxes drift)
OUT=gpurun_out/${1:-lds}; mkdir -p $OUT
for rep in 1 2; do
  for lib in armada_amd/csrc/libarmada_sched.so armada_amd/csrc/libarmada_sched_pad_*.so; do
    name=$(basename $lib .so); name=${name#libarmada_sched}; name=${name:-_default}
    ASCHED_LIB_PATH=$PWD/$lib timeout 300 python bench.py --steps 6 --warmup 2 --cpu-budget 0 --no-other > $OUT/b.json 2> $OUT/b.err
    python -c "import json;d=json.load(open('$OUT/b.json'));print('$name rep $rep', round(d['ms_per_step'],1), round(d['p50_ms'],1), 'k_control', round(d['round']['k_control_ms'],1))" | tee -a $OUT/summary.txt
  done
done
