"""smoke(): one tiny render of the Cornell box on cuda:0 checked against the oracle."""
import numpy as np


def run():
    import _harness as H
    import srt_amd
    from _cases import pt_sample_list, pt_scene

    scene = pt_scene("cbox")
    w = h = 32
    pt = srt_amd.Pathtracer(0)
    pt.set_params(w, h, 4, 8, True)
    pt.build_scene(scene)
    pt.set_camera(scene["camera"])
    img = pt.render_epoch(1, 0, 4)
    want = H.OraclePT(scene, w, h, 8, True).epoch(1, 0, 4)
    assert np.array_equal(img.view(np.uint32), want.view(np.uint32)), "path tracer smoke mismatch vs oracle"
    pt.set_elision(True)                                   # two-ray batches: same image, fewer rays traced
    pt.rays_elided(reset=True)
    assert np.array_equal(pt.render_epoch(1, 0, 4).view(np.uint32), want.view(np.uint32)), "elision changed the image"
    assert pt.rays_elided(reset=True) > 0
    pt.set_elision(False)
    t = np.load(H.GOLDEN + "/tonemap_render_32x32_e1.npz")  # display epilogue against the reference's bytes
    assert np.array_equal(pt.tonemap(t["rgb"], float(t["exposure"])), t["rgba"]), "tonemap differs from the reference golden"
    assert np.array_equal(pt.tonemap(img, 1.0), H.oracle_tonemap(img, 1.0))
    g = np.load(H.GOLDEN + "/pt_cbox_64x64_d8_bvh.npz")
    pt.set_params(64, 64, 1, 8, True)
    xs, ys, ss = pt_sample_list(int(g["seed"]), 64, 64, 4096)
    rgb, draws, _ = pt.trace_samples(int(g["seed"]), xs, ys, ss)
    assert np.array_equal(rgb.view(np.uint32), g["rgb"].view(np.uint32)) and np.array_equal(draws, g["draws"])
    pt.close()
    print("smoke: path tracer ok (bit-exact vs oracle and reference golden; elision and tonemap included)")
