#!/usr/bin/env bash
# sha256 of the disassembly of k_control (the round kernel) in a built library / object: two builds whose hashes agree run the same round-kernel code
#   tools/kcontrol_isa_hash.sh [path/to/libarmada_sched.so | armada_sched.o]
set -e
LIB=$(readlink -f "${1:-$(dirname "$0")/../armada_amd/csrc/libarmada_sched.so}")
T=$(mktemp -d); trap 'rm -rf $T' EXIT
BIN=/opt/rocm/lib/llvm/bin
cd "$T"
python3 - "$LIB" <<'PY'
import sys, re, struct
data = open(sys.argv[1], 'rb').read()
magic = b'__CLANG_OFFLOAD_BUNDLE__'
n = 0
for m in re.finditer(re.escape(magic), data):
    p = m.start()
    cnt = struct.unpack_from('<Q', data, p + 24)[0]
    off = p + 32
    for _ in range(cnt):
        o, s, tl = struct.unpack_from('<QQQ', data, off); off += 24
        triple = data[off:off + tl].decode(); off += tl
        if 'gfx950' in triple and s:
            open(f'co{n}.o', 'wb').write(data[p + o:p + o + s]); n += 1
PY
for f in co*.o; do
  if $BIN/llvm-readelf -s "$f" 2>/dev/null | grep -q " _Z9k_control3DeviP7HelpBoxi$"; then
    # the whole text of the round kernel's code object — k_control and the out-of-line functions it calls — as instruction text (addresses and encodings stripped)
    $BIN/llvm-objdump -d --no-show-raw-insn "$f" | grep -E "^\s+[a-z_0-9]+ " | sed -E 's/\/\/.*$//' > kc.txt
    echo "round-kernel code object: $(wc -l < kc.txt) instructions, sha256 $(sha256sum kc.txt | cut -c1-16)"
  fi
done
