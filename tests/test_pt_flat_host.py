"""The flattened per-lane walk (csrc/pt_flat.h) against the nested form (scene_hit, csrc/pt_trace.h) and the oracle,
on the CPU: the device headers are compiled for the host by tests/host_emu and run one lane at a time.  The HIP
build of the same source is checked by tests/test_pt_gpu.py (kernel mode 5)."""
import numpy as np
import pytest

import _harness as H
from _cases import pt_scene, random_rays


@pytest.mark.parametrize("name,use_bvh", [
    ("cbox", True), ("cbox", False), ("cbox_lambertian", True), ("cbox_blob512_glass", True), ("cbox_blob512_glass", False),
    ("cbox_blob2048_mirror", True), ("cbox_nolight", True),
])
def test_flat_walk_equals_nested_and_oracle(name, use_bvh):
    scene = pt_scene(name)
    emu = H.EmuPT(scene, use_bvh)
    org, dirs, bounds = random_rays(11, 3000)
    o = H.OraclePT(scene, 8, 8, 8, use_bvh).hit(org, dirs, bounds)          # {hit, dist, pos, normal, material}
    for slot in range(3):
        nested, flat = emu.hit(org, dirs, bounds, slot)
        assert np.array_equal(nested, flat), f"slot {slot}: flattened walk differs from the nested form"
    assert np.array_equal(nested[:, 0], o[:, 0].astype(np.uint32))
    assert np.array_equal(nested[:, 1][nested[:, 0] == 1], o[:, 1].view(np.uint32)[nested[:, 0] == 1]), "distance bits differ from the oracle"
    # three slots of one batch share the origin and are walked one after the other inside one loop
    n = 900
    d3 = dirs[:3 * n].reshape(n, 9)
    got = emu.hit3(org[:n], d3, bounds[:n])
    for s in range(3):
        want, _ = emu.hit(org[:n], np.ascontiguousarray(d3[:, 3 * s:3 * s + 3]), bounds[:n], 0)
        assert np.array_equal(got[:, s], want)
    emu.close()
