OUT=gpurun_out/r05y22; mkdir -p $OUT
for L in new nonest2; do
  P=$PWD/armada_amd/csrc/libarmada_sched_$L.so; [ $L = new ] && P=$PWD/armada_amd/csrc/libarmada_sched.so
  echo "== $L" >> $OUT/ab.txt
  ASCHED_LIB_PATH=$P timeout 300 python tools/dbg_nest.py 100345 102465 2>&1 | tail -n 2 | cut -c1-200 >> $OUT/ab.txt
  ASCHED_LIB_PATH=$P timeout 600 python -m pytest tests -q -m gpu -k "stream or gang or nest" -p no:cacheprovider 2>&1 | tail -n 2 >> $OUT/ab.txt
done
bash tools/ab_call.sh r05y22 "head nonest2 new" "headline gangs preempt"
