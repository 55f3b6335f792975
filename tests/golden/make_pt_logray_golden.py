#!/usr/bin/env python3
"""Golden vectors for Pathtracer::log_ray (rays/pathtracer.cpp:191-193 <- student/pathtracer.cpp:148) from the REFERENCE's own code.

Run in the authoring container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_pt_logray_golden.py

The reference build (oracle/_ref/libref_pt.so, RNG seam = SRT-RNG v1) renders one epoch of each case with the harness's sink
behind Gui::Widget_Render::log_ray; every call's arguments - ray.point, ray.dir, t, color - are recorded together with the
(pixel, sample) being traced and the call's ordinal within that sample.  Stored per case: the calls in call order (expected
outputs) and the epoch image.  `bounce` is the oracle's name for the ordinal (max_depth - ray depth at the shading point); the script
checks that the oracle logs the same rays in the same order before it writes anything.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _harness as H  # noqa: E402
from _cases import pt_scene, scene_digest  # noqa: E402

# scene, w, h, max_depth, use_bvh, samples per pixel, sample_base
CASES = [
    ("cbox", 64, 48, 8, True, 24, 5),                 # mirror + glass: discrete bounces log nothing, the ledger still moves
    ("cbox_lambertian", 48, 48, 8, True, 16, 0),      # the two-ray (elision) build's scene
    ("cbox_blob512_glass", 48, 40, 8, True, 16, 2),   # a real BVH<Triangle>: the streamed forms
    ("cbox_deltalights", 40, 32, 6, True, 12, 1),     # point_lighting in front of the coin
]
SEED = 20260404


def main():
    assert H.ref_pt_lib() is not None, "build oracle/_ref first: make -C oracle ref"
    for name, w, h, depth, use_bvh, spp, base in CASES:
        scene = pt_scene(name)
        ref = H.RefPT(scene, w, h, depth, use_bvh)
        img, log = ref.epoch_log(SEED, base, spp)
        assert len(log) > 8, f"{name}: only {len(log)} logged rays - raise the sample count"
        assert (log[:, 6] == 5.0).all() and (log[:, 10:13] == 1.0).all()      # log_ray(ray, 5.0f), color = Spectrum{1.0f}
        for mode in (0, 1):
            o = H.OraclePT(scene, w, h, depth, use_bvh, math_mode=mode)
            oimg, olog = o.epoch_log(SEED, base, spp)
            assert np.array_equal(oimg.view(np.uint32), img.view(np.uint32)) or (np.isnan(oimg) == np.isnan(img)).all()
            assert len(olog) == len(log), (name, len(olog), len(log))
            assert np.array_equal(olog[:, :9].view(np.uint32), log[:, :9].view(np.uint32)), f"{name}: the oracle logs other rays"
            # same order within a sample: bounces ascend as the ordinals do
            key = olog[:, 7].astype(np.int64) * (1 << 20) + olog[:, 8].astype(np.int64)
            for k in np.unique(key):
                m = key == k
                assert (np.diff(olog[m, 9]) > 0).all() and np.array_equal(log[m, 9], np.arange(m.sum(), dtype=np.float32))
        tag = f"ptlog_{name}_{w}x{h}_d{depth}_s{spp}"
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), meta=np.array([w, h, depth, int(use_bvh), spp, base], np.int64),
                            seed=np.array(SEED, np.uint64), scene=np.array(name), scene_sha256=np.array(scene_digest(scene)),
                            point=log[:, 0:3].copy(), dir=log[:, 3:6].copy(), t=log[:, 6].copy(), pixel=log[:, 7].astype(np.uint32),
                            sample=log[:, 8].astype(np.uint32), ordinal=log[:, 9].astype(np.uint32), bounce=olog[:, 9].astype(np.uint32),
                            color=log[:, 10:13].copy(), epoch=img)
        shading = float(len(log)) / 0.0005
        print(f"{tag}: {len(log)} log_ray calls (~{shading:.0f} coin flips), bounces {np.bincount(olog[:, 9].astype(int)).tolist()}")


if __name__ == "__main__":
    main()
