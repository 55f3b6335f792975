#!/bin/bash
# Per-kernel registers / scratch / occupancy of csrc/pt.hip (or the file given) as the compiler reports them - no GPU needed:
#   bash tools/kernel_resources.sh [file.hip] [filter-regex]
# (-Rpass-analysis=kernel-resource-usage; the object goes to /tmp).  What DESIGN.md's spill and occupancy figures are read from.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
src=${1:-pt.hip}
filter=${2:-pt_wave_kernel|pt_cast_kernel|raster_tiles|raster_bin|raster_setup}
cd "$root/soft-rendering-toolsets_amd/csrc"
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize --offload-arch=gfx950 -I../../include \
      -Wno-unused-function $EXTRA -Rpass-analysis=kernel-resource-usage -c "$src" -o /tmp/kres.o > /tmp/kres.log 2>&1 || { tail -20 /tmp/kres.log; exit 1; }
python3 - "$filter" <<'PY'
import re, subprocess, sys
txt = open('/tmp/kres.log').read()
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split('\n')[0].split(' [')[0]
    d = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"\(.*", "", d).replace("void srt::", "").replace("void (anonymous namespace)::", "")
    if not re.search(sys.argv[1], short):
        continue
    g = lambda k: (re.search(re.escape(k) + r": (\d+)", b) or [0, -1])[1]
    v, sc, oc, sg, ld = g('VGPRs'), g('ScratchSize [bytes/lane]'), g('Occupancy [waves/SIMD]'), g('SGPRs'), g('LDS Size [bytes/block]')
    print(f"{short:60s} vgpr {v:>4} scratch {sc:>5} B/lane  occupancy {oc}  sgpr {sg}  lds {ld}")
PY
