#!/usr/bin/env bash
# the product sources with extra -D switches -> armada_amd/csrc/libarmada_sched_<name>.so (A/B runs: ASCHED_LIB_PATH; tools/ab_call.sh)
#   tools/build_variant.sh eng0 -DENG_START_AFTER=0
set -e
NAME=$1; shift
cd "$(dirname "$0")/../armada_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing $*"
hipcc $F -c armada_sched.hip -o /tmp/armada_sched_$NAME.o &
hipcc $F -c armada_sched_aux.hip -o /tmp/armada_sched_aux_$NAME.o &
hipcc $F -c armada_sched_mgpu.hip -o /tmp/armada_sched_mgpu_$NAME.o &
hipcc $F -c armada_sched_ft.hip -o /tmp/armada_sched_ft_$NAME.o &
hipcc $F -c armada_sched_wk.hip -o /tmp/armada_sched_wk_$NAME.o &
wait
hipcc --offload-arch=gfx950 -fPIC -shared -pthread -o libarmada_sched_$NAME.so /tmp/armada_sched_$NAME.o /tmp/armada_sched_aux_$NAME.o /tmp/armada_sched_mgpu_$NAME.o /tmp/armada_sched_ft_$NAME.o /tmp/armada_sched_wk_$NAME.o
ls -la libarmada_sched_$NAME.so
