"""CPU: pin oracle/pt_oracle.c against the golden vectors produced by the reference's own path tracer
(tests/golden/make_pt_golden.py) and, in the authoring container, against the reference build itself.
Bar: bit-exact per-sample radiance (NaN == NaN), RNG draw counts, scene.hit records, BVH node arrays."""
import ctypes
import glob
import os

import numpy as np
import pytest

import _harness as H
from _cases import pt_sample_list, pt_scene, random_rays, scene_digest

GOLDENS = sorted(glob.glob(os.path.join(H.GOLDEN, "pt_*.npz")))


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


def load_case(path):
    g = np.load(path)
    w, h, depth, use_bvh, n = (int(x) for x in g["meta"])
    scene = pt_scene(str(g["scene"]))
    assert scene_digest(scene) == str(g["scene_sha256"]), "scene description drifted from the fixture"
    return g, scene, w, h, depth, bool(use_bvh), n, int(g["seed"])


def test_goldens_present():
    names = [os.path.basename(g) for g in GOLDENS]
    assert "pt_cbox_lambertian_64x64_d8_bvh.npz" in names and "pt_cbox_64x64_d8_bvh.npz" in names
    assert len(names) >= 7


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "srtmath"])
@pytest.mark.parametrize("path", GOLDENS, ids=[os.path.basename(g)[3:-4] for g in GOLDENS])
def test_oracle_matches_reference_golden(path, math_mode):
    g, scene, w, h, depth, use_bvh, n, seed = load_case(path)
    o = H.OraclePT(scene, w, h, depth, use_bvh, math_mode=math_mode)
    xs, ys, ss = pt_sample_list(seed, w, h, n)
    rgb, draws, rays = o.trace_samples(seed, xs, ys, ss)
    assert bits_equal(rgb, g["rgb"]), "per-sample radiance differs from the reference"
    assert np.array_equal(draws, g["draws"]), "RNG draw ledger differs from the reference"
    org, d, b = random_rays(seed + 1, 2048)
    assert bits_equal(o.hit(org, d, b), g["hits"]), "scene.hit differs from the reference"
    if use_bvh:
        boxes, links, order = o.dump_bvh(-1)
        assert bits_equal(boxes, g["tlas_boxes"]) and np.array_equal(links, g["tlas_links"])
        assert np.array_equal(order[: len(scene["objects"])], g["tlas_order"])
        for k in range(len(scene["objects"])):
            H.check_blas_against_golden(g, k, o.dump_bvh)
    if "epoch" in g:
        ew, eh, spp, base = (int(x) for x in g["epoch_meta"])
        o2 = H.OraclePT(scene, ew, eh, depth, use_bvh, math_mode=math_mode)
        assert bits_equal(o2.epoch(seed, base, spp), g["epoch"]), "epoch image differs from the reference"


LOG_GOLDENS = sorted(glob.glob(os.path.join(H.GOLDEN, "ptlog_*.npz")))


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "srtmath"])
@pytest.mark.parametrize("path", LOG_GOLDENS, ids=[os.path.basename(g)[6:-4] for g in LOG_GOLDENS])
def test_oracle_ray_log_matches_reference_golden(path, math_mode):
    """Pathtracer::log_ray: the rays the reference build handed to Gui::Widget_Render::log_ray over one epoch (arguments and call order,
    tests/golden/make_pt_logray_golden.py) against the oracle's - and, where the reference build is present, against it again."""
    g = np.load(path)
    w, h, depth, use_bvh, spp, base = (int(x) for x in g["meta"])
    scene = pt_scene(str(g["scene"]))
    assert scene_digest(scene) == str(g["scene_sha256"])
    o = H.OraclePT(scene, w, h, depth, bool(use_bvh), math_mode=math_mode)
    img, log = o.epoch_log(int(g["seed"]), base, spp)
    assert bits_equal(img, g["epoch"])
    assert len(log) == len(g["pixel"]) >= 8
    assert bits_equal(log[:, 0:3], g["point"]) and bits_equal(log[:, 3:6], g["dir"]) and bits_equal(log[:, 6], g["t"])
    assert np.array_equal(log[:, 7].astype(np.uint32), g["pixel"]) and np.array_equal(log[:, 8].astype(np.uint32), g["sample"])
    assert np.array_equal(log[:, 9].astype(np.uint32), g["bounce"])
    if H.have_reference() and math_mode == 0:
        rimg, rlog = H.RefPT(scene, w, h, depth, bool(use_bvh)).epoch_log(int(g["seed"]), base, spp)
        assert bits_equal(rimg, img) and bits_equal(rlog[:, :9], log[:, :9])
        assert (rlog[:, 10:13] == 1.0).all()            # color = Spectrum{1.0f}


def test_particle_step_matches_reference_golden():
    """Scene_Particles::Particle::update as restated by the oracle against the states the reference produced
    over three steps (tests/golden/make_particles_golden.py), NaN positions of the particles at rest included; and scene.hit
    for un-normalised directions with the default [0, inf] bounds, as the step sends them."""
    from _cases import particle_cloud, unnormalised_rays

    g = np.load(os.path.join(H.GOLDEN, "particles_cbox_blob512.npz"))
    scene = pt_scene(str(g["scene"]))
    assert scene_digest(scene) == str(g["scene_sha256"])
    seed, n, steps = (int(v) for v in g["meta"])
    o = H.OraclePT(scene, 8, 8, 8, True)
    pos, vel, age = particle_cloud(seed, n)
    for s in range(steps):
        pos, vel, age, alive = o.particles_update(pos, vel, age, float(g["dt"]), float(g["radius"]), max_iter=0)
        assert bits_equal(pos, g[f"pos{s}"]) and bits_equal(vel, g[f"vel{s}"]) and bits_equal(age, g[f"age{s}"]), f"step {s}"
        assert np.array_equal(alive, g[f"alive{s}"])
    org, d, b = unnormalised_rays(seed + 7, 1024)
    assert bits_equal(o.hit(org, d, b), g["hits_unnormalised"])
    if H.ref_pt_lib() is not None:     # authoring container: a second cloud straight against the reference build
        ref = H.RefPT(scene, 8, 8, 8, True)
        p2, v2, a2 = particle_cloud(seed + 1, 3000)
        want = ref.particles_update(p2, v2, a2, 0.02, 0.05)
        got = o.particles_update(p2, v2, a2, 0.02, 0.05, max_iter=0)
        assert all(bits_equal(x, y) for x, y in zip(got[:3], want[:3])) and np.array_equal(got[3], want[3])


def test_draw_ledger_cornell():
    """per Lambertian bounce 8 + 2 draws, 2 jitter draws per camera sample."""
    scene = pt_scene("cbox_lambertian")
    o = H.OraclePT(scene, 32, 32, 8, True)
    xs, ys, ss = pt_sample_list(1, 32, 32, 2000)
    _, draws, rays = o.trace_samples(1, xs, ys, ss)
    assert ((draws - 2) % 10 == 0).all()          # every shading event is Lambertian here
    bounces = (draws - 2) // 10
    assert (rays == 1 + 3 * bounces).all()        # camera ray + (BSDF-direct, MIS-direct, indirect) per event
    assert bounces.max() <= 8


def test_srtmath_matches_libm():
    """SRT-MATH v2 restates glibc's sinf/cosf: bit-identical on the argument ranges the renderer uses."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.random(1_000_000, dtype=np.float32) * np.float32(2 * np.pi),
                        rng.random(1_000_000, dtype=np.float32) * 2 - 1,
                        np.array([0.0, -0.0, 1e-5, 2 ** -13, 0.785398, 0.7853982, 1.5707964, 3.1415927, 6.2831855], np.float32)])
    c = np.zeros_like(x)
    s = np.zeros_like(x)
    H.oracle().srt_oracle_math_cos_sin(H.P(x), ctypes.c_size_t(len(x)), H.P(c), H.P(s))
    libm = ctypes.CDLL("libm.so.6")
    libm.cosf.restype = libm.sinf.restype = ctypes.c_float
    libm.cosf.argtypes = libm.sinf.argtypes = [ctypes.c_float]
    idx = rng.integers(0, len(x), 20000)
    idx[:9] = np.arange(len(x) - 9, len(x))
    for i in idx:
        assert c[i] == np.float32(libm.cosf(float(x[i]))) and s[i] == np.float32(libm.sinf(float(x[i]))), float(x[i])


def _atan2_args(seed, n):
    rng = np.random.default_rng(seed)
    y = np.concatenate([rng.random(n, dtype=np.float32) * 10, rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32),
                        np.array([0.0, -0.0, 1.0, 0.0, 5.0, np.inf, np.inf, 1e-30, 3.0, 2.0 ** 70], np.float32)])
    x = np.concatenate([(rng.random(n, dtype=np.float32) - 0.5) * 20, rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32),
                        np.array([1.0, -1.0, 1.0, 0.0, -0.0, np.inf, -np.inf, -1e30, 0.0, 1.0], np.float32)])
    return np.ascontiguousarray(y), np.ascontiguousarray(x)


def test_srt_math_atan2_is_glibc():
    """SRT-MATH v2's atan2f restates glibc 2.35's __ieee754_atan2f / __atanf (Spot_Light::sample): bit-identical to
    the host libm on the renderer's range, random bit patterns and the special cases."""
    y, x = _atan2_args(2, 400_000)
    out = np.zeros_like(y)
    H.oracle().srt_oracle_math_atan2(H.P(y), H.P(x), ctypes.c_size_t(len(y)), H.P(out))
    libm = ctypes.CDLL("libm.so.6")
    libm.atan2f.restype = ctypes.c_float
    libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
    idx = np.random.default_rng(1).integers(0, len(y), 40000)
    idx[:10] = np.arange(len(y) - 10, len(y))
    for i in idx:
        want = np.float32(libm.atan2f(float(y[i]), float(x[i])))
        assert out[i].view(np.uint32) == want.view(np.uint32) or (np.isnan(out[i]) and np.isnan(want)), (float(y[i]), float(x[i]))


def test_srt_math_acos_is_glibc():
    """SRT-MATH v2's acosf restates glibc 2.35's __ieee754_acosf (Samplers::Hemisphere::Uniform): bit-identical to libm."""
    x = np.concatenate([np.random.default_rng(4).random(200000, dtype=np.float32) * 2 - 1,
                        np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 0.49999997, 1e-20, 0.99999994, -0.99999994], np.float32)])
    out = np.zeros_like(x)
    H.oracle().srt_oracle_math_acos(H.P(x), ctypes.c_size_t(len(x)), H.P(out))
    libm = ctypes.CDLL("libm.so.6")
    libm.acosf.restype = ctypes.c_float
    libm.acosf.argtypes = [ctypes.c_float]
    idx = np.random.default_rng(5).integers(0, len(x), 30000)
    idx[:10] = np.arange(len(x) - 10, len(x))
    for i in idx:
        assert out[i].view(np.uint32) == np.float32(libm.acosf(float(x[i]))).view(np.uint32), float(x[i])


TONEMAP_GOLDENS = sorted(glob.glob(os.path.join(H.GOLDEN, "tonemap_*.npz")))


def test_srt_math_exp_pow_are_glibc():
    """expf / powf of SRT-MATH v2 (FMA build of glibc 2.35's algorithms) against this host's libm: a strided sweep of every
    exponent range of expf and the whole sRGB argument range of powf at y = 1/2.4 (the complete sweeps - every float for
    expf, every normal x at eight exponents for powf - were run when the restatement was written; see oracle/pt_oracle.c)."""
    lib = H.oracle()
    lib.srt_oracle_sweep_exp_vs_libm.restype = ctypes.c_uint64
    lib.srt_oracle_sweep_pow_vs_libm.restype = ctypes.c_uint64
    first = ctypes.c_uint32(0)
    bad = 0
    for start in range(0, 1 << 32, 1 << 26):                # 64 windows of 2^17 consecutive floats, all exponents, both signs
        bad += lib.srt_oracle_sweep_exp_vs_libm(ctypes.c_uint32(start), ctypes.c_uint32(start + (1 << 17)), ctypes.byref(first))
    bad += lib.srt_oracle_sweep_exp_vs_libm(ctypes.c_uint32(0xc2b00000), ctypes.c_uint32(0xc2d00000), ctypes.byref(first))   # -88 .. -104
    bad += lib.srt_oracle_sweep_exp_vs_libm(ctypes.c_uint32(0x4202422f), ctypes.c_uint32(0x42024230), ctypes.byref(first))   # unfused forms fail here
    bad += lib.srt_oracle_sweep_exp_vs_libm(ctypes.c_uint32(0xc27c65d9), ctypes.c_uint32(0xc27c65da), ctypes.byref(first))
    assert bad == 0, hex(first.value)
    lo = int(np.float32(0.0031308).view(np.uint32)) - 64
    hi = int(np.float32(1.0).view(np.uint32)) + 64
    y = ctypes.c_float(float(np.float32(1.0) / np.float32(2.4)))
    assert lib.srt_oracle_sweep_pow_vs_libm(ctypes.c_uint32(lo), ctypes.c_uint32(hi), y, ctypes.byref(first)) == 0, hex(first.value)
    for yy in (2.4, 5.0, 10.0, -1.5):
        assert lib.srt_oracle_sweep_pow_vs_libm(ctypes.c_uint32(0x3e000000), ctypes.c_uint32(0x3e000000 + (1 << 22)), ctypes.c_float(yy),
                                                ctypes.byref(first)) == 0, (yy, hex(first.value))


@pytest.mark.parametrize("path", TONEMAP_GOLDENS, ids=[os.path.basename(g)[8:-4] for g in TONEMAP_GOLDENS])
def test_oracle_tonemap_matches_reference_golden(path):
    g = np.load(path)
    assert np.array_equal(H.oracle_tonemap(g["rgb"], float(g["exposure"])), g["rgba"])


def test_tonemap_goldens_present():
    assert len(TONEMAP_GOLDENS) >= 4


@pytest.mark.skipif(H.ref_pt_lib() is None, reason="reference build (oracle/_ref) not present")
def test_oracle_tonemap_matches_reference_build():
    """Fresh seeded images, negative radiance and NaN included (their byte goes through the float -> unsigned char cast the
    way x86-64 compiles it), against the reference's HDR_Image::tonemap_to itself."""
    from _cases import tonemap_image

    for seed, exposure in ((101, 1.0), (102, 0.8), (103, 3.0)):
        rgb = tonemap_image("mixed", 96, 64, seed)
        rng = np.random.default_rng(seed)
        flat = rgb.reshape(-1)
        flat[rng.integers(0, flat.size, 200)] = np.float32(np.nan)
        flat[rng.integers(0, flat.size, 200)] = -np.abs(rng.normal(size=200)).astype(np.float32)
        assert np.array_equal(H.oracle_tonemap(rgb, exposure), H.ref_tonemap(rgb, exposure))


def test_accumulate_running_mean():
    rng = np.random.default_rng(3)
    acc = np.zeros(300, np.float32)
    want = np.zeros(300, np.float32)
    for k in range(1, 6):
        e = rng.random(300).astype(np.float32)
        H.oracle_accumulate(acc, e, k)
        want = (want + (e - want) * np.float32(1.0 / k)).astype(np.float32)
    assert np.array_equal(acc, want)


@pytest.mark.ref
@pytest.mark.skipif(H.ref_pt_lib() is None, reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("name,use_bvh", [("cbox", True), ("cbox_lambertian", False), ("cbox_blob512_glass", True)])
def test_oracle_matches_reference_build(name, use_bvh):
    scene = pt_scene(name)
    w, h = 40, 30
    ref = H.RefPT(scene, w, h, 8, use_bvh)
    xs, ys, ss = pt_sample_list(77, w, h, 6000, max_sample=1 << 20)
    r_rgb, r_draws = ref.trace_samples(123456789, xs, ys, ss)
    for mode in (0, 1):
        o = H.OraclePT(scene, w, h, 8, use_bvh, math_mode=mode)
        rgb, draws, _ = o.trace_samples(123456789, xs, ys, ss)
        assert bits_equal(rgb, r_rgb) and np.array_equal(draws, r_draws)
    assert H.ref_pt_lib().ref_pt_unqualified_sqrt_is_double() == 1  # shapes.cpp's sqrt(delta) is fp64


def test_fullsize_fixture_matches_scenes_and_oracle_crop():
    """tests/golden/pt_fullsize.json (full-size epoch hashes from the reference build): every case refers to the scene
    description as it is now, and the oracle reproduces the committed centre crop of the two cheapest cases bit for bit
    (the whole images are compared on the GPU, tests/test_pt_gpu.py::test_full_size_epoch_equals_reference_hash)."""
    cases = H.load_fullsize()
    assert {"cfg3_cbox_lambertian_512_64spp", "cfg4_cbox_1024_4spp", "cfg5_blob131072_1024_2spp", "cfg5_beast_1024_2spp"} <= set(cases)
    for name, g in cases.items():
        assert scene_digest(pt_scene(g["scene"])) == g["scene_sha256"], name
        assert len(g["sha256"]) == 64 and len(g["crop_hex"]) == 16 * 16 * 3 * 4 * 2
    for name in ("cfg4_cbox_1024_4spp", "cfg4_cbox_1024_3spp_base61"):
        g = cases[name]
        x0, y0 = g["crop_origin"]
        o = H.OraclePT(pt_scene(g["scene"]), g["w"], g["h"], g["max_depth"], True)
        img = np.zeros((g["h"], g["w"], 3), np.float32)
        o.epoch(g["seed"], g["sample_base"], g["spp"], y0, y0 + 16, img)
        crop = np.frombuffer(bytes.fromhex(g["crop_hex"]), np.float32).reshape(16, 16, 3)
        assert bits_equal(img[y0:y0 + 16, x0:x0 + 16], crop), name
