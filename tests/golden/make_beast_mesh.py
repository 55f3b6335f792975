#!/usr/bin/env python3
"""Extracts the geometry of the reference's largest mesh asset, Scotty3D/media/beast.dae (64 618 triangles), into a fixture:
positions (float32, as written in the file) and triangle vertex indices.  Data only - the COLLADA <float_array> of the
positions and the VERTEX column of <triangles><p>; run in the authoring container (reads /root/reference)."""
import os
import re
import sys

import numpy as np

SRC = "/root/reference/Assignments/Scotty3D/media/beast.dae"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mesh_beast.npz")


def main():
    text = open(SRC).read()
    pos = re.search(r'<float_array id="Beast-positions-array" count="(\d+)">([^<]*)</float_array>', text)
    positions = np.array(pos.group(2).split(), dtype=np.float32).reshape(-1, 3)
    assert positions.size == int(pos.group(1))
    tri = re.search(r'<triangles[^>]*count="(\d+)">(.*?)</triangles>', text, re.S)
    inputs = re.findall(r'<input semantic="(\w+)"[^>]*offset="(\d+)"', tri.group(2))
    stride = 1 + max(int(o) for _, o in inputs)
    voff = [int(o) for s, o in inputs if s == "VERTEX"][0]
    p = np.array(re.search(r"<p>([^<]*)</p>", tri.group(2)).group(1).split(), dtype=np.int64).reshape(-1, stride)
    triangles = p[:, voff].reshape(-1, 3).astype(np.int32)
    assert len(triangles) == int(tri.group(1))
    assert triangles.min() >= 0 and triangles.max() < len(positions)
    np.savez_compressed(OUT, positions=positions, triangles=triangles, source="Assignments/Scotty3D/media/beast.dae (geometry 'Beast')")
    print(OUT, positions.shape, triangles.shape, "bbox", positions.min(0), positions.max(0), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
