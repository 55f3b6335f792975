#!/usr/bin/env python3
"""Diagnostic: rasterizer frame time on the cfg2 stream (and a synthetic many-small-triangles stream)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
r = bench.raster_bench(0, frames=50, warmup=5)
print(json.dumps({k: r[k] for k in ("value", "ms_per_frame", "sample_tests_per_s", "bit_exact_vs_reference_golden")}))
