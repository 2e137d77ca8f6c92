#!/usr/bin/env bash
# `minsize` on the cold functions of the generic path gave a different (wrong) round on the preemption-heavy input.  Build one library per
# cold function with the attribute on that function only (-DCOLD_MINSIZE_MASK=1<<i, armada_amd/csrc/cold_attr.h) + one with all of them:
#   tools/minsize_bisect.sh build            (here: cross-compiles; 14 variants, 4 at a time)
#   gpurun -- 'bash tools/minsize_bisect.sh run'   (GPU box: one preemption-heavy round per variant, result fingerprint against the default build)
set -u
cd "$(dirname "$0")/.."
NAMES=(selectAtLevelLiteral selectWithFairPreemption fairApply selectAtPriority scheduleMany trySchedule gangSchedule replayEvicted updateFairShares pqsEvict ensureFairIndex ensureReplaySlow runRound)
case ${1:-build} in
build)
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing"
  cd armada_amd/csrc
  one() { local tag=$1 mask=$2; hipcc $F -DCOLD_MINSIZE_MASK=$mask -c armada_sched.hip -o /tmp/ms_$tag.o && hipcc --offload-arch=gfx950 -fPIC -shared -pthread -o libarmada_sched_ms_$tag.so /tmp/ms_$tag.o armada_sched_aux.o && echo built $tag; }
  N=0
  for i in "${!NAMES[@]}"; do one "${NAMES[$i]}" $((1 << i)) & N=$((N + 1)); if [ $((N % 4)) = 0 ]; then wait; fi; done
  one all 8191 &
  wait
  ls -la libarmada_sched_ms_*.so | wc -l ;;
run)
  OUT=gpurun_out/minsize; mkdir -p $OUT
  for L in default "${NAMES[@]}" all; do
    P=$PWD/armada_amd/csrc/libarmada_sched_ms_$L.so; [ $L = default ] && P=$PWD/armada_amd/csrc/libarmada_sched.so
    [ -f "$P" ] || continue
    echo "== $L" | tee -a $OUT/bisect.txt
    ASCHED_LIB_PATH=$P timeout 300 python tools/round_fingerprint.py 2>&1 | tail -n 2 | tee -a $OUT/bisect.txt
  done ;;
esac
