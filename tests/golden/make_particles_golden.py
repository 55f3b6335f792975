#!/usr/bin/env python3
"""Generate the particle-step golden fixture from the REFERENCE's own code (authoring container only; needs
`make -C oracle ref`):

    python tests/golden/make_particles_golden.py

The reference's Scene_Particles::Particle::update (student/particles.cpp, compiled into oracle/_ref/libref_pt.so) advances a
seeded particle cloud (tests/_cases.py:particle_cloud) through three 10 ms steps against the Cornell box with the glass blob
(a BVH<Object> over meshes, one with a real BVH<Triangle>, and a sphere); the fixture stores the state after every step, and
scene.hit records for rays with un-normalised directions and the default [0, inf] bounds, as the step sends them."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _harness as H  # noqa: E402
from _cases import particle_cloud, pt_scene, scene_digest, unnormalised_rays  # noqa: E402

SCENE, SEED, N, DT, RADIUS, STEPS = "cbox_blob512_glass", 314, 2048, 0.01, 0.02, 3


def main():
    assert H.ref_pt_lib() is not None, "build oracle/_ref first: make -C oracle ref"
    scene = pt_scene(SCENE)
    ref = H.RefPT(scene, 8, 8, 8, True)
    pos, vel, age = particle_cloud(SEED, N)
    out = dict(scene=np.array(SCENE), scene_sha256=np.array(scene_digest(scene)), meta=np.array([SEED, N, STEPS], np.int64),
               dt=np.float32(DT), radius=np.float32(RADIUS))
    for s in range(STEPS):
        pos, vel, age, alive = ref.particles_update(pos, vel, age, DT, RADIUS)
        out.update({f"pos{s}": pos, f"vel{s}": vel, f"age{s}": age, f"alive{s}": alive})
    org, d, b = unnormalised_rays(SEED + 7, 1024)
    out["hits_unnormalised"] = ref.hit(org, d, b)
    np.savez_compressed(os.path.join(HERE, "particles_cbox_blob512.npz"), **out)
    moved = np.abs(out[f"pos{STEPS - 1}"] - particle_cloud(SEED, N)[0]).max()
    print(f"particles: {N} x {STEPS} steps, alive after the last step {int(alive.sum())}, max displacement {moved:.3f}, "
          f"unnormalised hits {int(out['hits_unnormalised'][:, 0].sum())}/1024")


if __name__ == "__main__":
    main()
