#!/usr/bin/env bash
# Register / scratch / LDS use of every kernel in the shipped library (what the round-2 review extracted by hand):
#   tools/kernel_resources.sh [path/to/libarmada_sched.so]
set -e
LIB=$(readlink -f "${1:-$(dirname "$0")/../armada_amd/csrc/libarmada_sched.so}")
T=$(mktemp -d); trap 'rm -rf $T' EXIT
BIN=/opt/rocm/lib/llvm/bin
cd "$T"
$BIN/clang-offload-bundler --list --type=o --input="$LIB" >/dev/null 2>&1 || true
# the fat binary section holds one bundle per translation unit: pull out every gfx950 code object
python3 - "$LIB" <<'PY'
import sys, re, subprocess
data = open(sys.argv[1], 'rb').read()
magic = b'__CLANG_OFFLOAD_BUNDLE__'
pos = [m.start() for m in re.finditer(re.escape(magic), data)]
import struct
n = 0
for p in pos:
    cnt = struct.unpack_from('<Q', data, p + 24)[0]
    off = p + 32
    for _ in range(cnt):
        o, s, tl = struct.unpack_from('<QQQ', data, off); off += 24
        triple = data[off:off + tl].decode(); off += tl
        if 'gfx950' in triple and s:
            open(f'co{n}.o', 'wb').write(data[p + o:p + o + s]); n += 1
print(n, 'code objects')
PY
for f in co*.o; do
  $BIN/llvm-readelf --notes "$f" | awk '
    /\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_spill_count:/ {ss=$2} /\.vgpr_spill_count:/ {vs=$2}
    /\.private_segment_fixed_size:/ {sc=$2} /\.group_segment_fixed_size:/ {lds=$2} /\.sgpr_count:/ {s=$2}
    /\.wavefront_size:/ {printf "%-28s vgpr %4s sgpr %4s sgpr_spill %4s vgpr_spill %4s scratch %6s B  lds %7s B\n", name, v, s, ss, vs, sc, lds}'
done
