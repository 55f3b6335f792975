#!/bin/bash
# Registers / scratch / LDS of every kernel in a built object (default: the path tracer), read from the code object's notes.
set -e
OBJ=${1:-$(dirname "$0")/../soft-rendering-toolsets_amd/lib/pt.hip.o}
LLVM=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin "$OBJ"
$LLVM/clang-offload-bundler --unbundle --type=o --input=$TMP/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/k.hsaco
$LLVM/llvm-readelf --notes $TMP/k.hsaco | python3 -c "
import sys, re
cur = {}
def flush():
    if cur.get('.name'):
        print('%-70s vgpr %4s agpr %3s sgpr %4s scratch %5s lds %6s spill v/s %s/%s' % (cur['.name'][:70], cur.get('.vgpr_count'), cur.get('.agpr_count'), cur.get('.sgpr_count'),
              cur.get('.private_segment_fixed_size'), cur.get('.group_segment_fixed_size'), cur.get('.vgpr_spill_count'), cur.get('.sgpr_spill_count')))
for l in sys.stdin:
    m = re.match(r'\s*-?\s*(\.[a-z_]+):\s*(\S+)', l)
    if not m: continue
    k, v = m.groups()
    if k == '.agpr_count' and cur.get('.name') and '.agpr_count' in cur: pass
    if k in cur and k in ('.agpr_count',) : flush(); cur.clear()
    cur[k] = v
    if k == '.wavefront_size': flush(); cur.clear()
"
rm -rf $TMP
