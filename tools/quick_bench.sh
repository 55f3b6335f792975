#!/bin/bash
# short PT-only bench line: Mrays/s, ms/step, image hash, dominant-kernel ms
python bench.py --steps ${1:-8} --no-cpu-baseline --no-raster ${SRT_QB_ARGS:-} 2>&1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), 'Mrays/s', round(d['ms_per_step'], 2), 'ms/step', d['image_sha256_16'], 'kernel', round(d['roofline']['kernel_ms'], 2), 'ms')
e = d.get('dead_ray_elision')
if e: print('  elision:', round(e['ms_per_step'], 2), 'ms/step', round(e['reference_equivalent_mrays_per_s'], 1), 'ref-eq Mrays/s', round(e['traced_mrays_per_s'], 1), 'traced', round(e['rays_elided_fraction'], 3), 'elided', 'image equal:', e['image_equals_full_trace_bit_for_bit'])"
