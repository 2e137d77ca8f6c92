#!/usr/bin/env bash
# profiling build of the same sources: -DASCHED_FASTPROF puts shader-clock reads at the segment borders of the fast iteration (round_fast.h SEG())
# -> armada_amd/csrc/libarmada_sched_prof.so ; use: ASCHED_LIB_PATH=armada_amd/csrc/libarmada_sched_prof.so ASCHED_PRINT_SEG=1 python bench.py ...
set -e
cd "$(dirname "$0")/../armada_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -DASCHED_FASTPROF"
hipcc $F -c armada_sched.hip -o /tmp/armada_sched_prof.o &
hipcc $F -c armada_sched_aux.hip -o /tmp/armada_sched_aux_prof.o &
hipcc $F -c armada_sched_mgpu.hip -o /tmp/armada_sched_mgpu_prof.o &
hipcc $F -c armada_sched_ft.hip -o /tmp/armada_sched_ft_prof.o &
wait
hipcc --offload-arch=gfx950 -fPIC -shared -pthread -o libarmada_sched_prof.so /tmp/armada_sched_prof.o /tmp/armada_sched_aux_prof.o /tmp/armada_sched_mgpu_prof.o /tmp/armada_sched_ft_prof.o
ls -la libarmada_sched_prof.so
