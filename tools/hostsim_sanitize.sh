#!/usr/bin/env bash
# The device control code compiled for the CPU (tests/hostsim) under AddressSanitizer + UndefinedBehaviorSanitizer, over the first seeds of tests/soak.py's `rounds` and `streams`
# workloads (optionally in lag mode: HS_RING_LAG=<seed>).  Test infrastructure; nothing here is linked into the product.
#   tools/hostsim_sanitize.sh [rounds-seeds=200] [streams-seeds=25]
set -e
cd "$(dirname "$0")/.."
g++ -Itests/hostsim -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -Wno-unused-function -shared -pthread \
    -o /tmp/libhostsim_san.so tests/hostsim/hostsim.cpp
cat > /tmp/hostsim_san_run.py <<'PY'
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from armada_amd.binding import Library
from armada_amd import workloads as W
import test_z_stream_runs as T
hs = Library("/tmp/libhostsim_san.so", "asched_")
n = 0
for mk, cnt in ((T._soak_round_workload, int(sys.argv[1])), (T._soak_stream_workload, int(sys.argv[2]))):
    for seed in range(100000, 100000 + cnt):
        wl = mk(seed); s = W.load(hs, wl); W.prepare(s, wl); s.schedule_round(); s.close(); n += 1
print("rounds run under ASan + UBSan:", n)
PY
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 python /tmp/hostsim_san_run.py "${1:-200}" "${2:-25}"
