"""GPU: parity of the HIP rasterizer (through the C ABI) with the oracle and the reference goldens.
Bar: bit-exact RGBA8 and bit-exact float supersample buffer."""
import glob
import os

import numpy as np
import pytest

import _harness as H
from _cases import adversarial_stream, random_triangles

pytestmark = pytest.mark.gpu

GOLDENS = sorted(glob.glob(os.path.join(H.GOLDEN, "raster_*.npz")))


@pytest.fixture(scope="module")
def srt():
    import srt_amd

    return srt_amd


def render(srt, prims, w, h, sr, samples=False, textures=None):
    ren = srt.SoftwareRenderer(0)
    ren.set_render_target(None, w, h)
    ren.set_sample_rate(sr)
    for t in range(len(textures) if textures is not None else 0):
        assert ren.add_texture(textures.texture(t)) == t
    rgba = ren.draw_stream(prims).copy()
    ss = ren.read_samples() if samples else None
    st = ren.stats()
    ren.close()
    return rgba, ss, st


@pytest.mark.parametrize("path", GOLDENS, ids=[os.path.basename(g)[7:-4] for g in GOLDENS])
def test_hip_matches_reference_golden(srt, path):
    g = np.load(path)
    w, h, sr = (int(x) for x in g["meta"])
    rgba, ss, _ = render(srt, g["prims"], w, h, sr, samples=True, textures=H.Textures.from_npz(g))
    assert np.array_equal(rgba, g["rgba"]), f"{(rgba != g['rgba']).any(axis=2).sum()} pixels differ from the reference"
    assert H.sha(ss) == str(g["ss_sha256"]), "float supersample buffer differs from the reference"


@pytest.mark.parametrize("sr", [1, 2, 3, 4, 5, 8, 16, 32])
@pytest.mark.parametrize("wh", [(83, 59), (1, 1), (33, 7), (64, 64)])
def test_hip_matches_oracle_random(srt, sr, wh):
    w, h = wh
    prims = np.concatenate([random_triangles(17 * sr + w, 150, w, h, 40), adversarial_stream(sr + h, w, h)])
    rgba, ss, st = render(srt, prims, w, h, sr, samples=True)
    o_rgba, o_ss, c = H.oracle_raster_frame(prims, w, h, sr, want_samples=True)
    assert np.array_equal(rgba, o_rgba)
    assert np.array_equal(ss.view(np.uint32), o_ss.view(np.uint32))
    # work counters agree with the oracle's (same sample tests, same fragments)
    assert (st.sample_tests, st.sample_tests_in_target, st.fragments, st.point_samples) == tuple(int(x) for x in c)


def test_cfg2_counters(srt):
    """BASELINE configs[1]: test3.svg, 1024x1024, supersample 4 — the work the Mfrags/s metric counts."""
    g = np.load(os.path.join(H.GOLDEN, "raster_cfg2_test3_1024_ss4.npz"))
    _, _, st = render(srt, g["prims"], 1024, 1024, 4)
    assert st.sample_tests == 50000979
    assert abs(st.fragments - 3874153) <= 2


def test_empty_and_ragged(srt):
    rgba, ss, st = render(srt, np.zeros(0, H.PRIM_DTYPE), 5, 3, 2, samples=True)
    assert (rgba == 255).all() and (ss == 255.0).all() and st.fragments == 0
    # target not a multiple of the tile size, one pixel wide / tall
    for w, h, sr in ((1, 200, 3), (200, 1, 4), (31, 33, 1), (97, 61, 6)):
        prims = random_triangles(w * h, 60, w, h, 64)
        rgba, ss, _ = render(srt, prims, w, h, sr, samples=True)
        o_rgba, o_ss, _ = H.oracle_raster_frame(prims, w, h, sr, want_samples=True)
        assert np.array_equal(rgba, o_rgba) and np.array_equal(ss.view(np.uint32), o_ss.view(np.uint32))


def test_full_size_properties(srt):
    """Size-independent properties at the full cfg2 size with a large synthetic stream:
    (1) rendering is deterministic and idempotent (same stream -> same bytes, twice);
    (2) painter's order: an opaque target-covering triangle appended last yields a flat image;
    (3) splitting the stream across several submit calls changes nothing."""
    w = h = 1024
    sr = 4
    prims = random_triangles(99, 20000, w, h, 60)
    ren = srt.SoftwareRenderer(0)
    ren.set_render_target(None, w, h)
    ren.set_sample_rate(sr)
    a = ren.draw_stream(prims).copy()
    b = ren.resolve().copy()  # resolve again without clearing: same stream, same result
    assert np.array_equal(a, b)
    ren.clear_target()
    for part in np.array_split(prims, 7):
        ren.submit(part)
    c = ren.resolve().copy()
    assert np.array_equal(a, c)
    cover = random_triangles(1, 1, w, h, 1)
    cover["v"] = np.array([-5000, -5000, 9000, -5000, -5000, 9000], np.float32).view(np.float64)
    cover["rgba"] = [0.25, 0.5, 0.75, 1.0]
    d = ren.draw_stream(np.concatenate([prims, cover]))
    assert (d[..., :3] == [63, 127, 191]).all() and (d[..., 3] == 255).all()
    ren.close()
    # and the oracle agrees on a 1/16 crop of the same scene (full size takes the scalar oracle too long)
    sub = random_triangles(5, 3000, 256, 256, 60)
    g_rgba, _, _ = render(srt, sub, 256, 256, 4)
    o_rgba, _, _ = H.oracle_raster_frame(sub, 256, 256, 4)
    assert np.array_equal(g_rgba, o_rgba)


def test_error_behaviour(srt):
    ren = srt.SoftwareRenderer(0)
    with pytest.raises(srt.SrtError) as e:
        ren.submit(np.zeros(1, H.PRIM_DTYPE))  # before set_render_target
    assert e.value.status == -5
    with pytest.raises(srt.SrtError):
        ren.set_render_target(None, 0, 10)
    ren.set_render_target(None, 8, 8)
    with pytest.raises(srt.SrtError) as e:
        ren.set_sample_rate(33)  # tile kernel supports 1..32
    assert e.value.status == -4
    ren.sample_rate = 1
    bad = np.zeros(1, H.PRIM_DTYPE)
    bad["kind"] = 9
    with pytest.raises(srt.SrtError):
        ren.submit(bad)
    ren.close()


def test_many_small_triangles_and_big_target(srt):
    """200k small triangles into a 2048x1536 target at ss=2 (coarse binning with long lists); compared with the
    oracle on the full frame."""
    w, h, sr = 2048, 1536, 2
    prims = random_triangles(31337, 200000, w, h, 12, alpha=(0.3, 1.0))
    rgba, _, st = render(srt, prims, w, h, sr)
    o_rgba, _, c = H.oracle_raster_frame(prims, w, h, sr)
    assert np.array_equal(rgba, o_rgba)
    assert (st.sample_tests, st.fragments) == (int(c[0]), int(c[2]))
    assert st.list_bytes < 160 << 20, "the two levels of bin lists are sized by the frame's entries (8 bytes each: primitive + packed box), not bins x primitives"


def test_stress_svg_at_full_size_matches_reference(srt):
    """SURVEY.md 8(d)'s stress variant: hardcore/02_degenerate_square2.svg, 1024 x 1024, supersample 4 - 1000 frame-sized
    translucent triangles, 5.86 G sample tests, every coarse bin lists every primitive.  RGBA8 and the float sample buffer
    against the reference's (22 s there)."""
    g = np.load(os.path.join(H.GOLDEN, "stress_degenerate2_1024_ss4.npz"))
    w, h, sr = (int(x) for x in g["meta"])
    rgba, ss, st = render(srt, g["prims"], w, h, sr, samples=True)
    assert np.array_equal(rgba, g["rgba"])
    assert H.sha(ss) == str(g["ss_sha256"])
    assert st.sample_tests == 5858050128


def test_points_only_stream(srt):
    """Streams made only of rasterize_point records (what Xiaolin-Wu lines become), incl. heavy overdraw of one pixel."""
    import srt_amd

    rng = np.random.default_rng(4)
    w, h = 64, 48
    xy = np.floor(rng.random((5000, 2)) * [w, h])
    xy[:1500] = [10, 10]                      # 1500 translucent points on the same pixel: order matters
    rgba_in = rng.random((5000, 4)).astype(np.float32)
    prims = srt_amd.point_prims(xy, rgba_in)
    for sr in (1, 3, 4):
        got, ss, _ = render(srt, prims, w, h, sr, samples=True)
        want, o_ss, _ = H.oracle_raster_frame(prims, w, h, sr, want_samples=True)
        assert np.array_equal(got, want) and np.array_equal(ss.view(np.uint32), o_ss.view(np.uint32))


@pytest.mark.parametrize("sr", [1, 2, 3, 4, 8])
def test_images_match_oracle(srt, sr):
    """rasterize_image + Sampler2DImp::sample_trilinear on the GPU against the oracle: images over every border,
    minified and magnified, under and over translucent triangles; textures with the oracle's own mip chains;
    a second frame after srt_raster_clear_textures; an image record without its texture is an error."""
    from _cases import image_stream

    w, h = 75, 58
    prims, level0 = image_stream(900 + sr, w, h)
    tex = H.Textures.from_level0(level0, H.oracle_generate_mips)
    rgba, ss, _ = render(srt, prims, w, h, sr, samples=True, textures=tex)
    o_rgba, o_ss, _ = H.oracle_raster_frame(prims, w, h, sr, want_samples=True, textures=tex)
    assert np.array_equal(rgba, o_rgba)
    assert np.array_equal(ss.view(np.uint32), o_ss.view(np.uint32))

    ren = srt.SoftwareRenderer(0)
    ren.set_render_target(None, w, h)
    ren.set_sample_rate(sr)
    with pytest.raises(srt.SrtError):
        ren.draw_stream(prims)                      # no textures loaded
    for t in range(len(tex)):
        ren.add_texture(tex.texture(t))
    first = ren.draw_stream(prims).copy()
    assert np.array_equal(first, o_rgba)
    ren.clear_textures()
    only = H.Textures.from_level0(level0[:1], H.oracle_generate_mips)
    ren.add_texture(only.texture(0))
    sub = prims[(prims["kind"] != 3) | (prims["reserved"] == 0)]
    again = ren.draw_stream(sub).copy()
    want, _, _ = H.oracle_raster_frame(sub, w, h, sr, textures=only)
    assert np.array_equal(again, want)
    ren.close()


def test_textures_added_before_an_image_free_frame(srt):
    """add_texture, draw a stream WITHOUT image records, then one with them.  The first frame must not mark the
    textures as uploaded (it uploads nothing); the second samples the right texels.  Then the other orders around it: a texture
    set replaced between two image-free frames, and an image stream right after a retarget."""
    from _cases import image_stream

    w, h, sr = 75, 58, 2
    prims, level0 = image_stream(77, w, h)
    tex = H.Textures.from_level0(level0, H.oracle_generate_mips)
    tris = prims[prims["kind"] != 3]
    assert len(tris) and len(tris) < len(prims)
    ren = srt.SoftwareRenderer(0)
    ren.set_render_target(None, w, h)
    ren.set_sample_rate(sr)
    for t in range(len(tex)):
        ren.add_texture(tex.texture(t))
    assert np.array_equal(ren.draw_stream(tris), H.oracle_raster_frame(tris, w, h, sr)[0])
    assert ren.texture_upload_bytes() == 0
    want, _, _ = H.oracle_raster_frame(prims, w, h, sr, textures=tex)
    assert np.array_equal(ren.draw_stream(prims), want)
    assert ren.texture_upload_bytes() == int((4 * tex.level_w.astype(np.int64) * tex.level_h).sum())
    # another texture set (reversed order: other offsets, other texels at offset 0), an image-free frame, then images again
    rev = H.Textures.from_level0(level0[::-1], H.oracle_generate_mips)
    ren.clear_textures()
    for t in range(len(rev)):
        ren.add_texture(rev.texture(t))
    assert np.array_equal(ren.draw_stream(tris), H.oracle_raster_frame(tris, w, h, sr)[0])
    want_rev, _, _ = H.oracle_raster_frame(prims, w, h, sr, textures=rev)
    assert np.array_equal(ren.draw_stream(prims), want_rev)
    if len(level0) > 1:
        assert not np.array_equal(want_rev, want)
    # retarget: the image tables follow the target, the texels stay
    before = ren.texture_upload_bytes()
    ren.set_sample_rate(3)
    assert np.array_equal(ren.draw_stream(prims), H.oracle_raster_frame(prims, w, h, 3, textures=rev)[0])
    assert ren.texture_upload_bytes() == before
    ren.close()


def test_texture_residency_across_redraws(srt):
    """DrawSVG's redraw clears and re-adds every mip chain each frame.  Re-adding the same levels uploads
    nothing, and the unchanged image stream takes the identical-stream shortcut; one changed texel is seen."""
    g = np.load(os.path.join(H.GOLDEN, "raster_test7_image_256_ss2.npz"))        # basic/test7.svg, from the reference build
    w, h, sr = (int(x) for x in g["meta"])
    prims, tex, golden = g["prims"], H.Textures.from_npz(g), g["rgba"]
    assert (prims["kind"] == 3).any() and len(tex)
    ren = srt.SoftwareRenderer(0)
    ren.set_render_target(None, w, h)
    ren.set_sample_rate(sr)

    def redraw(t):
        ren.clear_textures()
        for k in range(len(t)):
            ren.add_texture(t.texture(k))
        return ren.draw_stream(prims).copy()

    assert np.array_equal(redraw(tex), golden)
    first = ren.texture_upload_bytes()
    assert first == int((4 * tex.level_w.astype(np.int64) * tex.level_h).sum())
    for _ in range(3):
        assert np.array_equal(redraw(tex), golden)
    assert ren.texture_upload_bytes() == first, "a redraw of unchanged textures uploaded texels"
    # one texel of the first level changes: it is seen, and the blob goes up again from that level on
    mod = H.Textures(tex.nlevels, tex.level_w, tex.level_h, tex.level_off, tex.blob.copy())
    mod.blob[int(mod.level_off[0]):int(mod.level_off[0]) + 4] ^= 0x5A
    want = H.oracle_raster_frame(prims, w, h, sr, textures=mod)[0]
    got = redraw(mod)
    assert np.array_equal(got, want)
    assert ren.texture_upload_bytes() > first
    assert np.array_equal(redraw(tex), golden)
    ren.close()


@pytest.mark.parametrize("sr", [1, 2, 3, 4, 5, 8, 16, 32])
@pytest.mark.parametrize("wh", [(83, 67), (9, 140), (150, 11), (256, 256)])
def test_lines_match_oracle(srt, sr, wh):
    """SRT_PRIM_LINE records: rasterize_line_xiaolinwu expanded on the device (raster_setup: end points, gradient, the serial
    intery chain; tile kernel: the fills) against the oracle's restatement of the reference loop - RGBA8, the float sample
    buffer and the number of in-bounds fill_sample calls."""
    from _cases import line_stream

    w, h = wh
    prims = line_stream(5000 + 31 * sr + w, w, h)
    rgba, ss, st = render(srt, prims, w, h, sr, samples=True)
    o_rgba, o_ss, c = H.oracle_raster_frame(prims, w, h, sr, want_samples=True)
    assert np.array_equal(rgba, o_rgba), f"{(rgba != o_rgba).any(axis=2).sum()} pixels differ"
    assert np.array_equal(ss.view(np.uint32), o_ss.view(np.uint32))
    assert (st.sample_tests, st.fragments, st.point_samples) == (int(c[0]), int(c[2]), int(c[3]))


def test_long_lines_and_growing_storage(srt):
    """Thousands of long lines on a 1024 x 768 target: the first frame's guesses for the line tables and the packed bin lists
    are too small, the library grows them and repeats the frame; a resubmission of the same stream, a different stream and the
    first one again all give the oracle's image (checked on a crop-sized twin, the scalar oracle is slow at full size)."""
    rng = np.random.default_rng(12)
    w, h, sr = 1024, 768, 2

    def lines(n, seed):
        r = np.random.default_rng(seed)
        p = np.zeros(n, H.PRIM_DTYPE)
        p["kind"] = 4
        v = np.zeros((n, 6), np.float32)
        v[:, 0:4] = r.random((n, 4)) * [w, h, w, h]
        p["v"] = v.view(np.float64).reshape(n, 3)
        p["rgba"][:, :3] = r.random((n, 3))
        return p

    a, b = lines(6000, 1), lines(9000, 2)
    ren = srt.SoftwareRenderer(0)
    ren.set_render_target(None, w, h)
    ren.set_sample_rate(sr)
    img_a = ren.draw_stream(a).copy()
    assert np.array_equal(ren.draw_stream(a), img_a)          # identical resubmission: nothing uploaded, nothing binned
    img_b = ren.draw_stream(b).copy()
    assert not np.array_equal(img_a, img_b)
    assert np.array_equal(ren.draw_stream(a), img_a)
    ren.close()
    # the same streams against the oracle at a size it finishes quickly (coordinates scaled: other lines, same code paths)
    for s, n in ((1, 700), (2, 900)):
        p = lines(n, s)
        v = p["v"].view(np.float32).reshape(n, 6) * np.float32(0.25)
        p["v"] = v.view(np.float64).reshape(n, 3)
        got, _, _ = render(srt, p, w // 4, h // 4, sr)
        want, _, _ = H.oracle_raster_frame(p, w // 4, h // 4, sr)
        assert np.array_equal(got, want)
    del rng


def test_frames_into_a_lent_framebuffer_equal_frames_into_a_fresh_one(srt):
    """srt_raster_resolve into the framebuffer srt_raster_bind_output pinned (what DrawSVG lends its renderer) against a fresh renderer
    that allocates its own pageable output: one renderer with a bound target draws frame after frame - every frame's pixels land on the
    previous image - widths that are odd or no multiple of anything, sample rates whose tiles are 32, 16, 10, 8 and 6 pixels wide, a
    target change in between, triangles, lines, points and images, up to 1920 x 1080.  (Round 4 built a read-back INSIDE the tile kernel
    on this path and removed it again - DESIGN.md section 1; this test is what caught its first hole, at 1920 x 1080.)"""
    from _cases import adversarial_stream, image_stream, line_stream, random_triangles

    frame = 0
    for w, h, sr in ((257, 130, 1), (512, 96, 2), (333, 77, 3), (1024, 64, 4), (96, 200, 5), (31, 9, 4), (640, 480, 1), (1024, 1024, 4), (1920, 1080, 2)):
        target = np.zeros((h, w, 4), np.uint8)
        ren = srt.SoftwareRenderer(0)
        ren.set_render_target(target, w, h)
        ren.set_sample_rate(sr)
        img, level0 = image_stream(900 + w, w, h)
        tex = H.Textures.from_level0(level0, H.oracle_generate_mips)
        for t in range(len(tex)):
            assert ren.add_texture(tex.texture(t)) == t
        for k in range(6):
            seed = 1000 * w + 10 * sr + k
            if k % 3 == 0:
                prims = random_triangles(seed, 40 + 25 * k, w, h, max(w, h) / 3)
            elif k % 3 == 1:
                prims = np.concatenate([line_stream(seed, w, h), adversarial_stream(seed, w, h)])
            else:
                prims = np.concatenate([random_triangles(seed, 30, w, h, max(w, h) / 2), img, line_stream(seed + 1, w, h)])
            got = ren.draw_stream(prims)
            assert got.ctypes.data == target.ctypes.data, "the frame is delivered into the lent framebuffer"
            want, _, _ = render(srt, prims, w, h, sr, textures=tex)
            assert np.array_equal(got, want), (w, h, sr, k)
            frame += 1
            if k == 3:                                            # the same stream again: the identical-stream shortcut, tiles + read-back only
                target[...] = 7
                assert np.array_equal(ren.draw_stream(prims), want)
        ren.close()
    assert frame == 54


def test_unwalkable_lines_are_refused(srt):
    """The reference's main loop `for (float x = xpxl1 + 1; x <= xpxl2 - sample_rate; ++x)` never ends when x reaches 2^24 or is
    infinite; such a line is refused by the product (SRT_ERR_UNSUPPORTED at resolve) and by the oracle alike.  A line with NaN
    coordinates draws nothing, one that merely ends far outside the target is drawn."""
    w, h, sr = 64, 48, 2

    def one(xy):
        p = np.zeros(1, H.PRIM_DTYPE)
        p["kind"] = 4
        v = np.zeros(6, np.float32); v[:4] = xy
        p["v"] = v.view(np.float64)
        p["rgba"] = [0.1, 0.2, 0.3, 1.0]
        return p

    for xy in ((5, 6, np.inf, 30), (-np.inf, 6, 20, 30), (5, np.inf, 20, 30), (5, 6, 3.0e7, 30), (-2.0e7, 1, 2.0e7, 40)):
        ren = srt.SoftwareRenderer(0)
        ren.set_render_target(None, w, h)
        ren.set_sample_rate(sr)
        with pytest.raises(srt.SrtError) as e:
            ren.draw_stream(one(xy))
        assert e.value.status == -4, xy
        ren.close()
        assert H.oracle().srt_oracle_raster_frame(H.P(one(xy)), 1, w, h, sr, H.P(np.zeros((h, w, 4), np.uint8)), None, None) == -1
    for xy in ((np.nan, 6, 20, 30), (5, 6, 20, np.nan), (-9.0e6, -20, 70, 60)):
        got, _, _ = render(srt, one(xy), w, h, sr)
        want, _, _ = H.oracle_raster_frame(one(xy), w, h, sr)
        assert np.array_equal(got, want)


def test_appending_and_retargeting_after_a_frame(srt):
    """The upload patches a line's ordinal / an image's table index into the pinned copy of the stream and puts the caller's words
    back only when the host next touches that copy (raster.hip: settle_upload).  Two sequences that depend on it: (1) submit,
    resolve, submit MORE without clearing, resolve - the second upload reads the first part again and needs the texture ids,
    not the table indices; (2) submit, resolve, change the sample rate, resolve - the same pinned stream is uploaded again."""
    from _cases import image_stream, line_stream

    w, h = 75, 58
    img, level0 = image_stream(4242, w, h)
    tex = H.Textures.from_level0(level0, H.oracle_generate_mips)
    first = np.concatenate([img, line_stream(77, w, h)])
    more = np.concatenate([line_stream(78, w, h), img[::-1]])
    ren = srt.SoftwareRenderer(0)
    ren.set_render_target(None, w, h)
    ren.set_sample_rate(2)
    for t in range(len(tex)):
        ren.add_texture(tex.texture(t))
    ren.clear_target()
    ren.submit(first)
    a = ren.resolve().copy()
    want_a, _, _ = H.oracle_raster_frame(first, w, h, 2, textures=tex)
    assert np.array_equal(a, want_a)
    ren.submit(more)                                  # (1) appended to the frame already drawn
    b = ren.resolve().copy()
    want_b, _, _ = H.oracle_raster_frame(np.concatenate([first, more]), w, h, 2, textures=tex)
    assert np.array_equal(b, want_b)
    ren.set_sample_rate(3)                            # (2) same stream, another sample rate
    c = ren.resolve().copy()
    want_c, _, _ = H.oracle_raster_frame(np.concatenate([first, more]), w, h, 3, textures=tex)
    assert np.array_equal(c, want_c)
    assert np.array_equal(ren.draw_stream(first), H.oracle_raster_frame(first, w, h, 3, textures=tex)[0])
    ren.close()
