#!/usr/bin/env python3
"""Generate the path-tracer golden fixtures from the REFERENCE's own code.

Run in the authoring container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_pt_golden.py

For each named scene (tests/_cases.py:pt_scene) the reference's PT::Pathtracer — compiled from its
sources by oracle/Makefile into oracle/_ref/libref_pt.so with clang++ (direct-before-indirect
evaluation order), RNG seam = SRT-RNG v1 — produces
  rgb, draws      trace_pixel radiance and RNG draw count for a list of (x, y, sample)   (expected outputs)
  hits            scene.hit records for a list of explicit rays                             (expected outputs)
  tlas_*, blas_*  BVH<Object> / BVH<Triangle> node arrays and primitive order               (expected outputs)
  epoch           one do_trace epoch image (mean of valid samples per pixel)                (expected outputs)
Inputs are regenerated from seeds by tests/_cases.py; scene_sha256 guards against drift of the scene
description itself.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _harness as H  # noqa: E402
from _cases import pt_sample_list, pt_scene, random_rays, scene_digest  # noqa: E402

# name, w, h, max_depth, use_bvh, n_samples, epoch (w, h, spp) or None
CASES = [
    ("cbox_lambertian", 64, 64, 8, True, 4096, (32, 32, 16)),   # BASELINE configs[2] geometry
    ("cbox", 64, 64, 8, True, 4096, (32, 32, 16)),              # BASELINE configs[3] geometry (mirror + glass)
    ("cbox", 48, 36, 3, False, 2048, None),                     # --no_bvh: List<Object> / List<Triangle>
    ("cbox_blob512_glass", 64, 64, 8, True, 4096, (24, 24, 8)), # BASELINE configs[4] stand-in, real BVH<Triangle>
    ("cbox_blob2048_mirror", 40, 40, 5, True, 2048, None),
    ("cbox_refract", 32, 32, 8, True, 1024, None),              # BSDF_Refract stub: zero direction -> NaN rays
    ("cbox_nolight", 32, 32, 4, True, 1024, (16, 16, 4)),       # empty area-light list -> every sample invalid
    ("cbox_deltalights", 48, 40, 8, True, 3072, (24, 20, 8)),   # point + spot + directional lights: point_lighting's shadow rays
    ("cbox_deltalights", 32, 24, 4, False, 1024, None),
    ("cbox_envsphere", 40, 32, 8, True, 2048, (20, 16, 8)),     # Env_Sphere + area light: coin-flipped sampling, mean of the pdfs
    ("cbox_envhemi", 32, 24, 5, True, 1024, None),              # Env_Hemisphere: radiance only for dir.y > 0
    ("cbox_envonly", 32, 24, 8, False, 1024, (16, 12, 4)),      # environment light alone (no area lights)
    ("cbox_envmap", 40, 32, 8, True, 2048, (20, 16, 6)),        # Env_Map: bilinear image lookup (acos / atan2), uniform sampling
    ("cbox_spherelight", 40, 32, 8, True, 2048, (20, 16, 6)),   # emissive analytic sphere lit through its triangle approximation
    # BASELINE configs[4] at size: 131 072-triangle glass mesh in the box, 1024 x 1024; the BVH<Triangle> (80 127 nodes) is
    # stored as SHA-256 digests of its arrays, the rest as for the other cases
    ("cbox_blob131072_glass", 1024, 1024, 8, True, 4096, (24, 24, 4)),
    # the reference's own largest mesh asset (media/beast.dae, 64 618 triangles) as the glass object, posed by a rotation + scale
    ("cbox_beast_glass", 512, 512, 8, True, 4096, (24, 24, 4)),
]
BIG_BLAS = 4096   # node arrays longer than this are stored as digests
SEED = 20260331


def main():
    assert H.ref_pt_lib() is not None, "build oracle/_ref first: make -C oracle ref"
    only = sys.argv[1:]
    for name, w, h, depth, use_bvh, n, ep in CASES:
        if only and name not in only:
            continue
        scene = pt_scene(name)
        ref = H.RefPT(scene, w, h, depth, use_bvh)
        xs, ys, ss = pt_sample_list(SEED, w, h, n)
        rgb, draws = ref.trace_samples(SEED, xs, ys, ss)
        org, d, b = random_rays(SEED + 1, 2048)
        hits = ref.hit(org, d, b)
        out = dict(meta=np.array([w, h, depth, int(use_bvh), n], np.int64), seed=np.array(SEED, np.uint64),
                   scene=np.array(name), scene_sha256=np.array(scene_digest(scene)), rgb=rgb, draws=draws, hits=hits)
        if use_bvh:
            boxes, links, order = ref.dump_bvh(-1)
            out.update(tlas_boxes=boxes, tlas_links=links, tlas_order=order[: len(scene["objects"])].copy())
            for k in range(len(scene["objects"])):
                d_ = ref.dump_bvh(k)
                if d_ is not None and len(d_[0]) > 1:
                    ntri = int(d_[1][0][1])
                    if len(d_[0]) > BIG_BLAS:
                        out.update({f"blas{k}_sha256": np.array([H.sha(d_[0]), H.sha(d_[1]), H.sha(d_[2][:ntri])]),
                                    f"blas{k}_nodes": np.array([len(d_[0]), ntri], np.int64)})
                    else:
                        out.update({f"blas{k}_boxes": d_[0], f"blas{k}_links": d_[1], f"blas{k}_order": d_[2][:ntri].copy()})
        if ep:
            ew, eh, spp = ep
            r2 = H.RefPT(scene, ew, eh, depth, use_bvh)
            out.update(epoch=r2.epoch(SEED, 3, spp), epoch_meta=np.array([ew, eh, spp, 3], np.int64))
        tag = f"pt_{name}_{w}x{h}_d{depth}_{'bvh' if use_bvh else 'list'}"
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
        finite = np.isfinite(rgb).all(axis=1)
        print(f"{tag}: {n} samples, {int(finite.sum())} finite, mean radiance {rgb[finite].mean():.4f}, "
              f"mean draws {draws.mean():.2f}, hits {int(hits[:, 0].sum())}/2048")


if __name__ == "__main__":
    main()
