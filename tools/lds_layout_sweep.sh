#!/usr/bin/env bash
# The round kernel keeps Dev and RoundScalars in LDS; growing Dev by 128 B + RoundScalars by 104 B cost the headline 3.5 % (profiles/r03l_*).  This builds variants of the
# SAME sources with padding at the end of either struct (armada_amd/csrc/libarmada_sched_<name>.so) to see what the layout does; run them with tools/lds_layout_run.sh on the GPU.
#   tools/lds_layout_sweep.sh            # builds every variant listed below, 3 at a time
set -e
cd "$(dirname "$0")/.."
VARIANTS="d128r104:-DASCHED_DEV_PAD=128,-DASCHED_RS_PAD=104 d128:-DASCHED_DEV_PAD=128 r104:-DASCHED_RS_PAD=104 d8:-DASCHED_DEV_PAD=8 d16:-DASCHED_DEV_PAD=16 d64:-DASCHED_DEV_PAD=64 d256:-DASCHED_DEV_PAD=256 r8:-DASCHED_RS_PAD=8 r16:-DASCHED_RS_PAD=16 r64:-DASCHED_RS_PAD=64 r256:-DASCHED_RS_PAD=256"
n=0
for v in $VARIANTS; do
  name=${v%%:*}; flags=$(echo "${v#*:}" | tr ',' ' ')
  ( tools/build_variant.sh pad_$name $flags > /tmp/build_pad_$name.log 2>&1 || echo "build of $name FAILED" ) &
  n=$((n+1)); if [ $((n % 3)) -eq 0 ]; then wait; fi
done
wait
ls -la armada_amd/csrc/libarmada_sched_pad_*.so
