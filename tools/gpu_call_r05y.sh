OUT=gpurun_out/r05z5; mkdir -p $OUT
timeout 100 python -m pytest tests -q -m gpu -k "stream or gang or nest or unfeasible" -p no:cacheprovider 2>&1 | tail -n 2 | tee $OUT/pytest_subset.txt
for L in prev new; do
  P=$PWD/armada_amd/csrc/libarmada_sched_$L.so; [ $L = new ] && P=$PWD/armada_amd/csrc/libarmada_sched.so
  echo "== $L gangsfull" | tee -a $OUT/ab.txt
  ASCHED_LIB_PATH=$P timeout 60 python tools/prof_config4.py gangsfull 2>&1 | grep "^round" | tail -n 1 | cut -c1-120 | tee -a $OUT/ab.txt
done
