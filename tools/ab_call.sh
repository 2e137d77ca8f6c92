#!/usr/bin/env bash
# A/B inside ONE GPU call (boxes differ by a few %): libraries armada_amd/csrc/libarmada_sched_<name>.so ("new" = the working tree's build)
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/ab_call.sh r02t "base new" "headline gangs preempt" "stream or gang"'
# workloads: headline | gangs | preempt | preempt_full | q256 | q1024 (tools/prof_config4.py); the optional pytest -k expression runs on the working tree's build afterwards.
# (round 4's one-off call scripts gpu_call_r4a … r4j were each an instance of this; their outputs are under profiles/r04*.)
set -u
export ASCHED_AB_OLD_LIB=1   # (armada_amd/binding.py: tolerate entry points an older build lacks)
TAG=${1:-ab}; LIBS=${2:-"base new"}; WHAT=${3:-"headline gangs preempt"}; KEXPR=${4:-}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
for rep in 1 2; do
  for L in $LIBS; do
    P=$PWD/armada_amd/csrc/libarmada_sched_$L.so; [ $L = new ] && P=$PWD/armada_amd/csrc/libarmada_sched.so
    unset ASCHED_EXCLUDED_NODES; [ $L = new_norec ] && { P=$PWD/armada_amd/csrc/libarmada_sched.so; export ASCHED_EXCLUDED_NODES=0; }   # the tree's build without the failed-selection records
    for W in $WHAT; do
      echo "== $L $W (rep $rep)" >> "$OUT/ab.txt"
      case $W in
        headline) ASCHED_LIB_PATH=$P timeout 600 python bench.py --steps 5 --warmup 1 --cpu-budget 0 --no-other 2>/dev/null | head -c 330 >> "$OUT/ab.txt"; echo >> "$OUT/ab.txt" ;;
        gangs) ASCHED_LIB_PATH=$P timeout 300 python tools/prof_config4.py gangs 2>&1 | tail -n 1 >> "$OUT/ab.txt" ;;
        preempt) [ $rep = 1 ] && ASCHED_LIB_PATH=$P timeout 300 python tools/prof_config4.py 2>&1 | grep "^round" | tail -n 1 | cut -c1-330 >> "$OUT/ab.txt" ;;
        preempt_full) [ $rep = 1 ] && ASCHED_LIB_PATH=$P timeout 400 python tools/prof_config4.py full 2>&1 | grep "^round" | tail -n 1 | cut -c1-330 >> "$OUT/ab.txt" ;;
        checker) ASCHED_LIB_PATH=$P timeout 300 python tools/prof_config4.py checker 2>&1 | grep "^round" | tail -n 1 | cut -c1-330 >> "$OUT/ab.txt" ;;
        q*) ASCHED_LIB_PATH=$P timeout 300 python tools/prof_config4.py $W 2>&1 | grep "^round" | tail -n 1 | cut -c1-330 >> "$OUT/ab.txt" ;;   # q256, q1024: wide runs
      esac
    done
  done
done
unset ASCHED_EXCLUDED_NODES ASCHED_AB_OLD_LIB
if [ -n "$KEXPR" ]; then timeout 900 python -m pytest tests -q -m gpu -k "$KEXPR" -p no:cacheprovider > "$OUT/pytest_ab.log" 2>&1; echo "pytest rc=$?" >> "$OUT/ab.txt"; tail -n 5 "$OUT/pytest_ab.log" >> "$OUT/ab.txt"; fi
cat "$OUT/ab.txt"
