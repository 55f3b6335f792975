#!/usr/bin/env python3
"""Full-size epoch goldens for the production kernels, from the REFERENCE's own code.

Run in the authoring container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_pt_fullsize_golden.py [name ...]

At BASELINE.json's image sizes an epoch image is megabytes, so what is committed per case is the SHA-256 of the
image's float32 bytes (row 0 = bottom, as do_trace leaves it), its mean over the finite pixels, the number of
non-finite pixels and a 16 x 16 crop (for a first look when a hash differs).  The image itself comes from
oracle/_ref/libref_pt.so = the reference's PT::Pathtracer::trace_pixel (clang++ -O2) with the seeded SRT-RNG, rows
spread over the host's threads (do_trace adds a pixel's samples in sample order, so the split does not matter).
tests/test_pt_gpu.py::test_full_size_epoch_equals_reference_hash renders the same epochs through the C ABI in `auto`
mode - the production kernels with their default population - and compares hashes: no reference build is needed on the
GPU box.
"""
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _harness as H  # noqa: E402
from _cases import pt_scene, scene_digest  # noqa: E402

# name, scene, w, h, max_depth, spp, seed, sample_base
CASES = [
    ("cfg3_cbox_lambertian_512_64spp", "cbox_lambertian", 512, 512, 8, 64, 0, 0),        # BASELINE configs[2], whole
    ("cfg4_cbox_1024_4spp", "cbox", 1024, 1024, 8, 4, 0, 0),                             # configs[3] at 4 of its 2048 spp
    ("cfg4_cbox_1024_3spp_base61", "cbox", 1024, 1024, 8, 3, 0, 61),                     # ... an odd count at a later sample index
    ("cfg4_cbox_1024_64spp", "cbox", 1024, 1024, 8, 64, 0, 0),                           # one whole step of bench.py (64 spp): the image every
                                                                                         # `bench.py --gpus N` / `--group N` run must reproduce
    ("cfg5_blob131072_1024_2spp", "cbox_blob131072_glass", 1024, 1024, 8, 2, 0, 0),      # configs[4] stand-in mesh
    ("cfg5_beast_1024_2spp", "cbox_beast_glass", 1024, 1024, 8, 2, 0, 0),                # configs[4] with the reference's largest asset
]
OUT = os.path.join(HERE, "pt_fullsize.json")


def render(scene, w, h, depth, seed, base, spp, threads):
    ref = H.RefPT(scene, w, h, depth, True)
    img = np.zeros((h, w, 3), np.float32)
    bounds = np.linspace(0, h, threads * 4 + 1).astype(int)
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda k: ref.epoch_rows(seed, base, spp, int(bounds[k]), int(bounds[k + 1]), img), range(len(bounds) - 1)))
    return img


def main():
    assert H.ref_pt_lib() is not None, "build oracle/_ref first: make -C oracle ref"
    only = sys.argv[1:]
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    threads = os.cpu_count() or 1
    for name, scene_name, w, h, depth, spp, seed, base in CASES:
        if only and name not in only:
            continue
        scene = pt_scene(scene_name)
        t0 = time.time()
        img = render(scene, w, h, depth, seed, base, spp, threads)
        fin = np.isfinite(img).all(axis=2)
        out[name] = dict(scene=scene_name, scene_sha256=scene_digest(scene), w=w, h=h, max_depth=depth, spp=spp, seed=seed, sample_base=base,
                         sha256=hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest(),
                         mean_finite=float(img[fin].mean(dtype=np.float64)), nonfinite_pixels=int((~fin).sum()),
                         crop_origin=[w // 2 - 8, h // 2 - 8],
                         crop_hex=np.ascontiguousarray(img[h // 2 - 8:h // 2 + 8, w // 2 - 8:w // 2 + 8]).tobytes().hex())
        print(f"{name}: {w}x{h} x {spp} spp in {time.time() - t0:.1f} s on {threads} threads, sha256 {out[name]['sha256'][:16]}, "
              f"mean {out[name]['mean_finite']:.5f}, non-finite pixels {out[name]['nonfinite_pixels']}", flush=True)
        json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
