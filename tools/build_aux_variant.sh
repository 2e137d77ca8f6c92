#!/usr/bin/env bash
# the auxiliary kernel's translation unit alone with other optimisation flags, linked with the product's round-kernel and multi-GPU objects
# -> armada_amd/csrc/libarmada_sched_<name>.so (the round kernel k_control is byte-identical to the product's: only k_control_aux differs)
#   tools/build_aux_variant.sh auxOs -Os
set -e
NAME=$1; shift
cd "$(dirname "$0")/../armada_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing "$@" -c armada_sched_aux.hip -o /tmp/armada_sched_aux_$NAME.o
hipcc --offload-arch=gfx950 -fPIC -shared -pthread -o libarmada_sched_$NAME.so armada_sched.o /tmp/armada_sched_aux_$NAME.o armada_sched_mgpu.o
ls -la libarmada_sched_$NAME.so
