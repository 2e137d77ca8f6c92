#!/usr/bin/env bash
# fair index rebuild period: configs[4] reduced / full
set -u
OUT=gpurun_out/r3w; mkdir -p "$OUT"
export TMPDIR=/tmp
for v in fr128 default512 fr2048 fr8192; do
  if [ $v = default512 ]; then unset ASCHED_LIB_PATH; else export ASCHED_LIB_PATH=armada_amd/csrc/libarmada_sched_$v.so; fi
  for shape in reduced full; do
    timeout 400 python tools/prof_config4.py $shape 2>&1 | grep "^round" | tail -1 | cut -c1-240 | sed "s/^/$v $shape /" | tee -a "$OUT/summary.txt"
  done
done
