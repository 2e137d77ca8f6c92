#!/usr/bin/env bash
# per-function comparison of the round kernel's code object in two builds: which functions' instruction text differs
#   tools/kcontrol_func_diff.sh libA.so libB.so [function-to-diff]
set -e
A=$(readlink -f "$1"); B=$(readlink -f "$2"); FN=${3:-}
T=$(mktemp -d); trap 'cp $T/a.dis $T/b.dis ${KEEP_DIS:-$T}/ 2>/dev/null; rm -rf $T' EXIT
BIN=/opt/rocm/lib/llvm/bin
extract() {  # $1 lib, $2 out prefix
python3 - "$1" "$2" <<'PY'
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
            open(f'{sys.argv[2]}{n}.o', 'wb').write(data[p + o:p + o + s]); n += 1
PY
}
cd "$T"
extract "$A" a; extract "$B" b
for side in a b; do
  for f in $side*.o; do
    if $BIN/llvm-readelf -s "$f" 2>/dev/null | grep -q " _Z9k_control3DeviP7HelpBoxi$"; then
      $BIN/llvm-objdump -d --no-show-raw-insn "$f" | sed -E 's/\/\/.*$//' > $side.dis
    fi
  done
done
python3 - "$FN" <<'PY'
import re, sys, hashlib, difflib
def funcs(path):
    out = {}; cur = None
    for line in open(path):
        m = re.match(r'^[0-9a-f]+ <(.+)>:', line)
        if m: cur = m.group(1); out[cur] = []; continue
        if cur and re.match(r'^\s+[a-z_0-9]+ ', line):
            t = line.strip()
            t = re.sub(r'\s+', ' ', t)
            out[cur].append(t)
    return out
def norm(ins):  # branch targets and pc-relative literals move with the layout: compare opcode + register operands only for the summary
    return [re.sub(r'(s_c?branch\S*|s_call\S*) .*', r'\1', re.sub(r'0x[0-9a-f]+|-?\b\d+\b', 'N', i)) for i in ins]
a, b = funcs('a.dis'), funcs('b.dis')
fn = sys.argv[1]
for name in sorted(set(a) | set(b)):
    if name not in a or name not in b: print('only in one build:', name); continue
    if norm(a[name]) != norm(b[name]):
        print(f'DIFF {name}: {len(a[name])} vs {len(b[name])} instructions')
        if fn and fn in name:
            for l in difflib.unified_diff(norm(a[name]), norm(b[name]), lineterm='', n=4): print(l)
print(f'{len(a)} functions compared')
PY
