"""Synthetic input generators shared by the golden-fixture scripts and the tests."""
import numpy as np

from _harness import PRIM_DTYPE


def _tri(p, xy, rgba):
    p["kind"] = 1
    p["v"] = np.asarray(xy, np.float32).reshape(6).view(np.float64)
    p["rgba"] = np.asarray(rgba, np.float32)


def _pt(p, xy, rgba):
    p["kind"] = 2
    p["v"][:2] = np.asarray(xy, np.float64)
    p["rgba"] = np.asarray(rgba, np.float32)


def adversarial_stream(seed, w, h):
    """Ordered stream of triangles and points that exercises the edge cases of
    rasterize_triangle / inside_triangle / fill_sample / rasterize_point:
    both windings, vertices exactly on sample corners, zero-area and coincident-vertex triangles,
    off-target and target-covering boxes, sub-sample triangles, 1e-20-scale triangles whose fp32
    sign products are denormal, translucent / out-of-gamut colours (clamp), fractional and negative
    point coordinates (double -> int truncation folds two block samples onto row/column 0)."""
    rng = np.random.default_rng(seed)
    prims = []

    def color():
        k = rng.integers(0, 6)
        a = [0.0, 0.25, 0.5, 1.0, 0.75, 1.0][k]
        c = rng.random(3)
        if rng.random() < 0.15:
            c = c * 2.5 - 0.7  # out of gamut: exercises both clamp ends
        return [c[0], c[1], c[2], a]

    def add_tri(xy):
        p = np.zeros((), PRIM_DTYPE)
        _tri(p, xy, color())
        prims.append(p)

    def add_pt(xy):
        p = np.zeros((), PRIM_DTYPE)
        _pt(p, xy, color())
        prims.append(p)

    # generic triangles, vertices on a 1/12 grid so that edges pass exactly through sample corners
    # for every sample_rate in 1..4
    for _ in range(60):
        v = rng.integers(-12 * 8, 12 * (max(w, h) + 8), size=6) / 12.0
        v[0::2] = np.clip(v[0::2], -8, w + 8)
        v[1::2] = np.clip(v[1::2], -8, h + 8)
        add_tri(v)
    # arbitrary float vertices, small triangles
    for _ in range(60):
        c = rng.random(2) * [w, h]
        v = (c[None, :] + (rng.random((3, 2)) - 0.5) * rng.choice([0.3, 2.0, 9.0, 40.0])).reshape(6)
        add_tri(v)
    # both windings of the same triangle, translucent, overlapping
    base = np.array([10.25, 7.5, 60.0, 12.0, 33.5, 50.75])
    add_tri(base)
    add_tri(base.reshape(3, 2)[::-1].reshape(6))
    # degenerate: collinear, two equal vertices, three equal vertices (on and off a sample corner)
    add_tri([5, 5, 20, 20, 40, 40])
    add_tri([5, 30, 25, 30, 45, 30])
    add_tri([7, 3, 7, 3, 30, 9])
    add_tri([12, 12, 12, 12, 12, 12])
    add_tri([12.3, 40.6, 12.3, 40.6, 12.3, 40.6])
    # covers everything / far outside / straddling each border
    add_tri([-1000, -1000, 3000, -1000, -1000, 3000])
    add_tri([-500, -500, -400, -480, -450, -300])
    add_tri([w + 100, 5, w + 200, 50, w + 150, 90])
    add_tri([-20, h / 2, 15, h / 2 - 9, 15, h / 2 + 9])
    add_tri([w - 10, -30, w + 30, 20, w - 25, 25])
    add_tri([30, h - 5, 60, h + 40, 10, h + 20])
    # 1e-20-scale triangles at sample corners: the pairwise cross products are ~1e-40 (fp32 denormal);
    # flushing them to zero would flip the coverage verdict of the neighbouring samples
    for s in (1e-19, 1e-20, 3e-21, 1e-22):
        for org in ((0.0, 0.0),):
            ox, oy = org
            add_tri([ox, oy, ox + s, oy, ox, oy + s])
            add_tri([ox, oy, ox, oy + s, ox + s, oy])
            add_tri([ox + s, oy + s, ox + 2 * s, oy + s, ox + s, oy + 3 * s])
    # thin slivers
    for _ in range(10):
        a = rng.random(2) * [w, h]
        d = (rng.random(2) - 0.5) * 80
        add_tri([a[0], a[1], a[0] + d[0], a[1] + d[1], a[0] + d[0] + 1e-3, a[1] + d[1] - 1e-3])
    # points: integer (Wu lines produce these), fractional, negative fractional (fold), far away
    for _ in range(40):
        add_pt(np.floor(rng.random(2) * [w + 4, h + 4]) - 2)
    for _ in range(30):
        add_pt(rng.random(2) * [w + 4, h + 4] - 2)
    for xy in ((-0.5, 3.0), (4.0, -0.25), (-0.75, -0.1), (-0.999, 10.5), (-1.0, 5.0), (-1.25, 6.0),
               (0.0, 0.0), (w - 1, h - 1), (w - 0.5, h - 0.5), (w, h), (1e9, 5.0), (5.0, -1e12), (3e9, 3e9)):
        add_pt(xy)
    # a second layer of translucent triangles on top of everything (painter's order matters)
    for _ in range(25):
        v = rng.random(6) * np.tile([w, h], 3)
        add_tri(v)
    out = np.stack(prims).astype(PRIM_DTYPE)
    # shuffle lightly so that points and triangles interleave, then keep that order as THE order
    idx = np.arange(len(out))
    rng.shuffle(idx)
    return out[idx].copy()


def _line(p, xy, rgba):
    p["kind"] = 4
    v = np.zeros(6, np.float32)
    v[:4] = np.asarray(xy, np.float32)
    p["v"] = v.view(np.float64)
    p["rgba"] = np.asarray(rgba, np.float32)


def line_stream(seed, w, h):
    """Ordered stream that exercises rasterize_line_xiaolinwu (software_renderer.cpp:365-454) and its interplay with the other
    primitives: every octant, steep / shallow, |slope| exactly 1 and 0, vertical and horizontal lines, end points on pixel
    centres / corners / halves (round() ties), lines shorter than a pixel and of zero length (both end points on one pixel:
    the four end-point fills overlap), lines shorter than the sample rate (the main loop's bound subtracts it), lines that
    leave the target on every side or lie entirely outside, long lines across the whole target, huge off-target end points,
    NaN coordinates (nothing must be drawn, nothing may hang), translucent and out-of-gamut colours (the stroke
    alpha is replaced by the coverage), under and over translucent triangles and points (painter's order)."""
    rng = np.random.default_rng(seed)
    prims = []

    def color():
        c = rng.random(3)
        if rng.random() < 0.15:
            c = c * 2.5 - 0.7
        return [c[0], c[1], c[2], [0.0, 0.3, 1.0, 0.6][rng.integers(0, 4)]]

    def add(kind, xy):
        p = np.zeros((), PRIM_DTYPE)
        (_line if kind == 4 else _tri if kind == 1 else _pt)(p, xy, color())
        prims.append(p)

    # a layer of translucent triangles underneath
    for _ in range(12):
        add(1, rng.random(6) * np.tile([w, h], 3))
    # random segments of every length and direction, end points on a 1/4 grid (round() ties, exact halves) and arbitrary
    for k in range(70):
        a = rng.random(2) * [w + 16, h + 16] - 8
        ln = rng.choice([0.3, 1.0, 2.5, 6.0, 20.0, 90.0])
        ang = rng.random() * 2 * np.pi
        b = a + ln * np.array([np.cos(ang), np.sin(ang)])
        if k % 3 == 0:
            a, b = np.round(a * 4) / 4, np.round(b * 4) / 4
        add(4, [a[0], a[1], b[0], b[1]])
    # exact slopes: horizontal, vertical, +-1 (dx == dy: `steep` is false, the tie goes to the shallow branch)
    for d in ((25, 0), (0, 25), (25, 25), (25, -25), (-25, 25), (-25, -25), (-25, 0), (0, -25), (1, 1), (3, 3), (0.5, 0.5)):
        a = np.floor(rng.random(2) * [w - 30, h - 30]) + 15 + rng.choice([0.0, 0.5, 0.25])
        add(4, [a[0], a[1], a[0] + d[0], a[1] + d[1]])
    # zero length and sub-pixel: the four end-point fills land on one or two pixels
    for xy in ((10, 10, 10, 10), (20.5, 7.5, 20.5, 7.5), (30.25, 9.75, 30.3, 9.8), (5.5, 30.5, 5.9, 30.6), (40.49, 12.0, 40.51, 12.0),
               (12.0, 40.49, 12.0, 40.51)):
        add(4, xy)
    # lengths around the sample rate (the main loop runs from xpxl1 + 1 to xpxl2 - sample_rate)
    for L in (1, 2, 3, 4, 5, 6, 7):
        y = 3 + 2 * L
        add(4, [50.0, y, 50.0 + L, y + 0.3 * L])
        add(4, [y + 0.5, 45.0, y + 0.8, 45.0 + L])
    # across the target, through its corners and borders, and entirely outside
    add(4, [-30, -20, w + 25, h + 35]); add(4, [w + 10, -10, -10, h + 10]); add(4, [-5, h / 2, w + 5, h / 2 + 0.7])
    add(4, [w / 2 + 0.2, -9, w / 2 - 0.4, h + 9]); add(4, [0, 0, w, h]); add(4, [0, h - 1, w - 1, 0])
    add(4, [-50, -50, -10, -3]); add(4, [w + 5, 5, w + 60, 40]); add(4, [5, h + 3, 60, h + 3]); add(4, [-0.4, 3, -0.4, 30]); add(4, [3, -0.6, 40, -0.6])
    add(4, [w - 0.5, 2, w - 0.5, h - 2]); add(4, [2, h - 0.5, w - 2, h - 0.5]); add(4, [w - 1, 0, w - 1, h]); add(4, [0, h - 1, w, h - 1])
    # far end points (long main loops, mostly off target) and non-finite coordinates
    add(4, [-3000.5, 10, w + 10, 40]); add(4, [20, -4000, 60, h + 5]); add(4, [-1e5, -1e5, w / 2, h / 2]); add(4, [w / 2, h / 2, 2e5, 1.5e5])
    # (infinite coordinates are not here: the reference's main loop never ends on them - `++x` on an infinite float - and the
    #  product refuses such lines, tests/test_raster_gpu.py::test_unwalkable_lines_are_refused)
    add(4, [np.nan, 5, 20, 30]); add(4, [5, np.nan, 20, 30]); add(4, [5, 6, np.nan, 30]); add(4, [5, 6, 20, np.nan])
    add(4, [np.nan, np.nan, np.nan, np.nan])
    # points and triangles in between and on top
    for _ in range(15):
        add(2, np.floor(rng.random(2) * [w, h]))
    for _ in range(10):
        add(1, rng.random(6) * np.tile([w, h], 3))
    out = np.stack(prims).astype(PRIM_DTYPE)
    idx = np.arange(len(out))
    rng.shuffle(idx)
    return out[idx].copy()


def random_triangles(seed, n, w, h, max_extent, alpha=(0.2, 1.0)):
    """n random triangles (for size-independent property tests and the benchmark's synthetic mode)."""
    rng = np.random.default_rng(seed)
    p = np.zeros(n, PRIM_DTYPE)
    c = rng.random((n, 1, 2)) * [w, h]
    v = (c + (rng.random((n, 3, 2)) - 0.5) * max_extent).astype(np.float32).reshape(n, 6)
    p["kind"] = 1
    p["v"] = v.view(np.float64).reshape(n, 3)
    p["rgba"][:, :3] = rng.random((n, 3))
    p["rgba"][:, 3] = alpha[0] + rng.random(n) * (alpha[1] - alpha[0])
    return p


# ------------------------------------------------------------------------------------------------
# path-tracer cases
# ------------------------------------------------------------------------------------------------
def pt_scene(name):
    """Named scenes used by the golden fixtures (deterministic: same name -> same arrays)."""
    import srt_amd  # noqa: F401  (loads the package under its importable name)
    from soft_rendering_toolsets_amd import scenes

    if name in ("cbox", "cbox_lambertian"):
        return scenes.cornell_box(name)
    if name == "cbox_blob512_glass":
        return scenes.cornell_with_mesh(3, "glass")   # 8*4^3 = 512 triangles -> a real BVH<Triangle>
    if name == "cbox_blob2048_mirror":
        return scenes.cornell_with_mesh(4, "mirror")
    if name == "cbox_blob131072_glass":
        # BASELINE configs[4] stand-in at size: Cornell walls + light + mirror sphere + a 131 072-triangle glass mesh
        # (the Stanford dragon is a missing large blob of the reference checkout; DESIGN.md names the substitute)
        return scenes.cornell_with_mesh(7, "glass")
    if name == "cbox_beast_glass":
        # the reference's own largest mesh asset (S/media/beast.dae, 64 618 triangles; geometry extracted by
        # tests/golden/make_beast_mesh.py) as a glass object in the Cornell box, Z-up -> Y-up, turned 30 degrees, scaled 0.22
        import os
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mesh_beast.npz"))
        T = [0.19052559, -0.11, 0, 0.18, 0, 0, 0.22, 0.016, -0.11, -0.19052559, 0, 0.2, 0, 0, 0, 1]
        return scenes.cornell_with_asset(g["positions"], g["triangles"], T, "glass", "beast")
    if name == "cbox_refract":
        s = scenes.cornell_box("cbox")
        s["materials"][6] = {"type": scenes.REFRACT, "a": np.ones(3, np.float32), "b": np.zeros(3, np.float32), "ior": 1.5}
        return s
    if name == "cbox_deltalights":
        # the Cornell box (area light kept) plus one light of each Delta_Light kind: a posed point light, a spot light
        # looking down with a smoothstep cone, a tilted directional light; the discrete BSDFs skip point_lighting
        s = scenes.cornell_box("cbox")

        def pose(t, rx=0.0, rz=0.0):
            cx, sx, cz, sz = np.cos(rx), np.sin(rx), np.cos(rz), np.sin(rz)
            Rx = np.array([[1, 0, 0, 0], [0, cx, -sx, 0], [0, sx, cx, 0], [0, 0, 0, 1]], np.float32)
            Rz = np.array([[cz, -sz, 0, 0], [sz, cz, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
            Tm = np.eye(4, dtype=np.float32)
            Tm[:3, 3] = t
            return np.ascontiguousarray((Tm @ Rz @ Rx).T.astype(np.float32).reshape(16))   # column-major

        s["lights"] = [
            {"type": 1, "radiance": np.array([0.6, 0.5, 0.4], np.float32), "T": pose((0.3, 0.55, 0.2))},
            {"type": 2, "radiance": np.array([2.0, 2.0, 2.5], np.float32), "angle_bounds": np.array([35.0, 70.0], np.float32),
             "T": pose((-0.25, 0.9, -0.1), rx=0.35, rz=-0.2)},
            {"type": 0, "radiance": np.array([0.15, 0.2, 0.15], np.float32), "T": pose((0.0, 0.0, 0.0), rx=0.5, rz=0.3)},
            {"type": 1, "radiance": np.array([0.2, 0.2, 0.2], np.float32), "T": np.eye(4, dtype=np.float32).reshape(16)},   # at the origin, no transform
        ]
        return s
    if name == "cbox_particles":
        # what Scene_Particles turns into: the Cornell box plus 60 small posed copies of one 32-triangle mesh and a few
        # extra spheres - a BVH<Object> of 74 objects, more than the wave-uniform kernels take
        s = scenes.cornell_box("cbox_lambertian")
        rng = np.random.default_rng(21)
        v, f = scenes.blob_mesh(1, seed=3, radius=1.0)
        flat_pos = v[f.reshape(-1)].astype(np.float32)
        tri_n = np.cross(flat_pos[1::3] - flat_pos[0::3], flat_pos[2::3] - flat_pos[0::3])
        tri_n /= np.linalg.norm(tri_n, axis=1, keepdims=True)
        flat_nrm = np.repeat(tri_n, 3, axis=0).astype(np.float32)
        idx = np.arange(len(flat_pos), dtype=np.uint32)
        s["materials"].append({"type": scenes.LAMBERTIAN, "a": np.array([0.2, 0.6, 0.3], np.float32), "b": np.zeros(3, np.float32), "ior": 1.0})
        mat = len(s["materials"]) - 1
        for k in range(60):
            Tm = np.eye(4, dtype=np.float32)
            Tm[0, 0] = Tm[1, 1] = Tm[2, 2] = np.float32(0.03)                      # Mat4::translate(p.pos) * Mat4::scale(scale)
            Tm[:3, 3] = (rng.random(3) * [1.2, 0.8, 1.2] - [0.6, -0.1, 0.6]).astype(np.float32)
            s["objects"].append({"kind": "mesh", "pos": flat_pos, "nrm": flat_nrm, "idx": idx, "T": np.ascontiguousarray(Tm.T.reshape(16)),
                                 "material": mat, "is_light": False})
        for k in range(6):
            Tm = np.eye(4, dtype=np.float32)
            Tm[:3, 3] = (rng.random(3) * [1.0, 0.6, 1.0] - [0.5, -0.2, 0.5]).astype(np.float32)
            s["objects"].append({"kind": "sphere", "radius": 0.04, "T": np.ascontiguousarray(Tm.T.reshape(16)), "material": mat, "is_light": False})
        return s
    if name == "cbox_spherelight":
        # an emissive analytic sphere next to the quad light: the sphere is intersected analytically, its triangle
        # approximation (an octahedron subdivided twice, 128 triangles, scaled to the radius) is what the area-light
        # sampler and pdf see (rays/pathtracer.cpp:105-131)
        s = scenes.cornell_box("cbox")
        radius = 0.12
        v, f = scenes.blob_mesh(2, seed=1, radius=1.0)
        v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
        flat_pos = (v[f.reshape(-1)] * np.float32(radius)).astype(np.float32)          # unindexed, flat-shaded like GL meshes
        tri_n = np.cross(flat_pos[1::3] - flat_pos[0::3], flat_pos[2::3] - flat_pos[0::3])
        tri_n /= np.linalg.norm(tri_n, axis=1, keepdims=True)
        flat_nrm = np.repeat(tri_n, 3, axis=0).astype(np.float32)
        idx = np.arange(len(flat_pos), dtype=np.uint32)
        s["materials"].append({"type": scenes.DIFFUSE_LIGHT, "a": np.array([4.0, 3.0, 2.0], np.float32), "b": np.zeros(3, np.float32), "ior": 1.0})
        Tm = np.eye(4, dtype=np.float32)
        Tm[:3, 3] = (-0.3, 0.45, 0.25)
        s["objects"].append({"kind": "sphere", "radius": radius, "T": np.ascontiguousarray(Tm.T.reshape(16)), "material": len(s["materials"]) - 1,
                             "is_light": True, "light_mesh": {"pos": flat_pos, "nrm": flat_nrm, "idx": idx}})
        return s
    if name == "cbox_envmap":
        # the open Cornell box under an image environment map (Env_Map): a smooth sky gradient with a bright sun blob
        # and some per-texel noise, 37 x 19 so that neither dimension is a power of two
        s = pt_scene("cbox_envsphere")
        rng = np.random.default_rng(5)
        hh, ww = 19, 37
        yy, xx = np.mgrid[0:hh, 0:ww].astype(np.float32)
        img = np.stack([0.2 + 0.5 * yy / hh, 0.3 + 0.4 * xx / ww, 0.9 - 0.5 * yy / hh], -1).astype(np.float32)
        img += (6.0 * np.exp(-((xx - 9) ** 2 + (yy - 4) ** 2) / 6.0))[..., None].astype(np.float32)
        img += rng.random((hh, ww, 3), dtype=np.float32) * 0.1
        s["env"] = {"type": 3, "image": np.ascontiguousarray(img, np.float32)}
        return s
    if name in ("cbox_envsphere", "cbox_envhemi", "cbox_envonly"):
        # an open Cornell box (no ceiling, no back wall) under an environment light, so that rays escape:
        # uniform sphere + the area light (coin-flipped sampling, mean of the pdfs); hemisphere + the area light;
        # hemisphere alone (sample_area_lights always samples the environment)
        s = scenes.cornell_box("cbox")
        keep = [o for o in s["objects"] if not (o["kind"] == "mesh" and not o["is_light"] and
                                                 (np.asarray(o["T"], np.float32).reshape(4, 4).T @ np.array([0, 0, 0, 1], np.float32))[1] > 0.95)]
        s["objects"] = keep[:1] + keep[2:] if len(keep) == len(s["objects"]) else keep   # make sure something is removed
        if name == "cbox_envonly":
            s["objects"] = [o for o in s["objects"] if not (o["kind"] == "mesh" and o["is_light"])]
        s["env"] = {"type": 1 if name == "cbox_envsphere" else 2, "radiance": np.array([0.7, 0.8, 1.0], np.float32)}
        return s
    if name in ("lone_blob_env", "lone_blob_env_lambertian"):
        # ONE object, a 512-triangle mesh with a real BVH<Triangle>, under a uniform sphere light: the BVH<Object> root is a
        # leaf (no interior node to sweep), every ray either enters the mesh's tree or leaves into the environment
        s = scenes.cornell_with_mesh(3, "glass")
        s["objects"] = [s["objects"][6]]
        if name.endswith("lambertian"):
            s["objects"][0] = dict(s["objects"][0], material=2)
        s["env"] = {"type": 1, "radiance": np.array([0.7, 0.8, 1.0], np.float32)}
        return s
    if name == "cbox_nolight":
        s = scenes.cornell_box("cbox_lambertian")
        s["objects"] = s["objects"][:-1]   # no area light: sample_area_lights returns the zero vector -> NaN rays
        return s
    raise KeyError(name)


def scene_digest(scene):
    import hashlib

    h = hashlib.sha256()
    for m in scene["materials"]:
        h.update(np.asarray([m["type"]], np.int32).tobytes() + np.asarray(m["a"], np.float32).tobytes()
                 + np.asarray(m["b"], np.float32).tobytes() + np.asarray([m["ior"]], np.float32).tobytes())
    for o in scene["objects"]:
        h.update(np.asarray(o["T"], np.float32).tobytes() + np.asarray([o["material"]], np.int32).tobytes())
        if o["kind"] == "mesh":
            h.update(np.asarray(o["pos"], np.float32).tobytes() + np.asarray(o["nrm"], np.float32).tobytes()
                     + np.asarray(o["idx"], np.uint32).tobytes() + bytes([int(o["is_light"])]))
        else:
            h.update(np.asarray([o["radius"]], np.float32).tobytes())
            if o.get("light_mesh") is not None:
                lm = o["light_mesh"]
                h.update(np.asarray(lm["pos"], np.float32).tobytes() + np.asarray(lm["nrm"], np.float32).tobytes() + np.asarray(lm["idx"], np.uint32).tobytes())
    if scene.get("env"):
        e = scene["env"]
        h.update(np.asarray([e["type"]], np.int32).tobytes() + np.asarray(e["image"] if int(e["type"]) == 3 else e["radiance"], np.float32).tobytes())
    for l in scene.get("lights", []):
        h.update(np.asarray([l["type"]], np.int32).tobytes() + np.asarray(l["radiance"], np.float32).tobytes()
                 + np.asarray(l.get("angle_bounds", (0.0, 0.0)), np.float32).tobytes() + np.asarray(l["T"], np.float32).tobytes())
    c = scene["camera"]
    h.update(np.asarray(c["iview"], np.float32).tobytes() + np.asarray([c["vfov"], c["ar"]], np.float32).tobytes())
    return h.hexdigest()


def pt_sample_list(seed, w, h, n, max_sample=4096):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, w, n).astype(np.uint32), rng.integers(0, h, n).astype(np.uint32),
            rng.integers(0, max_sample, n).astype(np.uint32))


def random_rays(seed, n):
    """Rays from inside and outside the Cornell box, some axis-aligned (zero direction components make
    1/dir = +-inf in BBox::hit), some with tight distance bounds."""
    rng = np.random.default_rng(seed)
    org = (rng.random((n, 3)) * [1.6, 1.4, 2.2] - [0.8, 0.2, 0.8]).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    k = n // 10
    d[:k] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, k)] * rng.choice([-1.0, 1.0], (k, 1)).astype(np.float32)
    d[k:2 * k, rng.integers(0, 3)] = 0.0
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    b = np.zeros((n, 2), np.float32)
    b[:, 0] = np.where(rng.random(n) < 0.5, 0.0, 1e-5)
    b[:, 1] = np.where(rng.random(n) < 0.3, rng.random(n) * 1.5, np.finfo(np.float32).max)
    return org, d.astype(np.float32), b


def particle_cloud(seed, n):
    """Particles as Scene_Particles emits them (scene/particles.cpp:143-159), scattered through the Cornell box: speeds up to the
    default 25 units / s (several bounces per 10 ms step near a wall), some slow ones, some resting on the floor or sitting
    closer to a wall than their radius (negative hit_time), some about to die.  Returns (pos, vel, age)."""
    rng = np.random.default_rng(seed)
    pos = (rng.random((n, 3)) * [0.9, 0.9, 0.9] + [-0.45, 0.05, -0.45]).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    speed = np.exp(rng.uniform(np.log(0.05), np.log(25.0), n))
    vel = (d * speed[:, None]).astype(np.float32)
    k = n // 16
    pos[:k, 1] = np.float32(0.011)                                 # just above the floor, moving down: hit inside the radius
    vel[:k, 1] = -np.abs(vel[:k, 1])
    vel[k:2 * k] = 0.0                                             # at rest: the ray has a zero direction
    vel[2 * k:3 * k, 0] = 0.0; vel[2 * k:3 * k, 2] = 0.0           # straight up / down (axis-aligned: 1/dir = inf in BBox::hit)
    age = rng.uniform(-0.005, 0.5, n).astype(np.float32)
    return pos, vel, age


def unnormalised_rays(seed, n):
    """Rays as Particle::update sends them: Ray() default bounds [0, inf] and dir = a velocity (norm 0.01 .. 30)."""
    org, d, b = random_rays(seed, n)
    rng = np.random.default_rng(seed + 1)
    d = (d * np.exp(rng.uniform(np.log(0.01), np.log(30.0), (n, 1)))).astype(np.float32)
    b = np.zeros((n, 2), np.float32)
    b[:, 1] = np.inf
    return org, d, b


def image_stream(seed, w, h):
    """(prims, [level-0 textures]) with image records among translucent triangles: a big magnified image, images
    hanging over every border (negative coordinates fold two loop values onto sample column / row 0), one minified
    to a few samples (last mip level), one stretched, a zero-width one, non-square and odd-sized textures."""
    rng = np.random.default_rng(seed)
    texs = [
        rng.integers(0, 256, (16, 16, 4), dtype=np.uint8),
        rng.integers(0, 256, (8, 32, 4), dtype=np.uint8),     # h = 8, w = 32
        rng.integers(0, 256, (13, 7, 4), dtype=np.uint8),     # odd sizes: levels round down
        np.full((4, 4, 4), 255, np.uint8),
    ]
    texs[0][..., 3] = 255                                     # one opaque texture
    rects = np.array([
        [5.25, 4.5, 70.75, 60.25],        # magnified
        [-7.5, -3.25, 20.5, 18.0],        # over the top-left corner: folds on column 0 and row 0
        [w - 12.5, h - 9.75, w + 15.0, h + 6.0],   # over the bottom-right corner
        [30.0, 10.0, 33.0, 12.5],         # minified: a 16x16 texture on 3x2.5 pixels
        [10.0, 40.0, 80.0, 44.0],         # stretched
        [50.0, 50.0, 50.0, 60.0],         # zero width: u = +inf -> clamped
        [-0.5, 30.0, 6.5, 36.0],          # starts at -0.5
        [40.0, 20.0, 72.0, 52.0],         # 1:1-ish for the 32-wide texture
    ], np.float32)
    ids = np.array([0, 1, 2, 0, 1, 3, 2, 1], np.uint32)
    img = np.zeros(len(rects), PRIM_DTYPE)
    img["kind"] = 3
    img["reserved"] = ids
    v = np.zeros((len(rects), 6), np.float32)
    v[:, :4] = rects
    img["v"] = v.view(np.float64).reshape(-1, 3)
    tri = random_triangles(seed + 1, 12, w, h, 50, alpha=(0.3, 0.9))
    prims = np.concatenate([tri[:4], img[:3], tri[4:8], img[3:6], tri[8:], img[6:]])
    return prims, texs


def tonemap_image(name, w, h, seed):
    """Radiance images for the tone-mapping tests: 'render' = an epoch image of the Cornell box out of the path-tracer
    golden; 'mixed' = seeded values over every branch of HDR_Image::tonemap_to / Spectrum::to_srgb: zeros, values below
    and around the linear/gamma knee (0.0031308), mid-range, saturating, infinities."""
    if name == "render":
        import os
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pt_cbox_64x64_d8_bvh.npz"))
        ep = np.asarray(g["epoch"], np.float32)
        assert ep.shape[:2] == (h, w), ep.shape
        return np.ascontiguousarray(ep[..., :3])
    rng = np.random.default_rng(seed)
    n = w * h * 3
    v = np.exp(rng.uniform(np.log(1e-6), np.log(60.0), n)).astype(np.float32)
    knee = np.float32(0.0031308)
    k = rng.integers(0, n, n // 16)
    v[k] = (knee * (1.0 + rng.uniform(-1e-3, 1e-3, len(k)))).astype(np.float32)     # 1 - exp(-x) ~ x near the knee
    v[rng.integers(0, n, n // 32)] = 0.0
    v[rng.integers(0, n, n // 64)] = np.float32(np.inf)
    v[rng.integers(0, n, n // 64)] = np.float32(200.0)
    v[rng.integers(0, n, n // 64)] = np.float32(1e-30)
    return np.ascontiguousarray(v.reshape(h, w, 3))


def random_pt_scene(seed):
    """Seeded random scene for differential tests of the kernels: the Cornell walls and light with jittered (no longer axis
    aligned) poses and random albedos, plus up to seven extra objects - spheres, small meshes (one BVH leaf) and
    subdivided blobs (a real BVH<Triangle>) - with random rotations / scales / translations and Lambertian / mirror / glass
    materials.  Returns (scene, width, height, max_depth, use_bvh, spp)."""
    import srt_amd  # noqa: F401  (loads the package under its importable name)
    from soft_rendering_toolsets_amd import scenes as sc

    rng = np.random.default_rng(seed)
    base = sc.cornell_box("cbox")

    def rot(ax, ang):
        c, s = np.cos(ang), np.sin(ang)
        m = np.eye(4)
        i, j = [(1, 2), (0, 2), (0, 1)][ax]
        m[i, i] = c; m[j, j] = c; m[i, j] = -s; m[j, i] = s
        return m

    def pose(scale, t, wobble):
        m = np.eye(4)
        for ax in range(3):
            m = m @ rot(ax, rng.uniform(-wobble, wobble))
        m[:3, :3] *= scale
        m[:3, 3] = t
        return m

    mats = []
    def material():
        kind = rng.choice([0, 0, 0, 1, 2])
        if kind == 0:
            mats.append(sc._mat(sc.LAMBERTIAN, sc.to_linear(rng.uniform(0.1, 0.95, 3))))
        elif kind == 1:
            mats.append(sc._mat(sc.MIRROR, rng.uniform(0.6, 1.0, 3)))
        else:
            mats.append(sc._mat(sc.GLASS, rng.uniform(0.7, 1.0, 3), rng.uniform(0.7, 1.0, 3), rng.uniform(1.2, 1.8)))
        return len(mats) - 1

    objs = []
    for k in range(5):                                         # walls: the file's matrix times a small wobble
        o = dict(base["objects"][k])
        T = np.asarray(o["T"], np.float64).reshape(4, 4).T @ pose(rng.uniform(0.98, 1.1), rng.uniform(-0.01, 0.01, 3), 0.02 if seed % 3 else 0.0)
        o["T"] = sc._colmajor(T.reshape(-1))
        mats.append(sc._mat(sc.LAMBERTIAN, sc.to_linear(rng.uniform(0.2, 0.9, 3))))
        o["material"] = len(mats) - 1
        objs.append(o)
    for _ in range(int(rng.integers(0, 8))):
        t = rng.uniform([-0.35, 0.1, -0.35], [0.35, 0.7, 0.35])
        kind = rng.integers(0, 4)
        if kind == 0:
            objs.append({"kind": "sphere", "radius": float(rng.uniform(0.05, 0.22)), "T": sc._colmajor(pose(rng.uniform(0.7, 1.3), t, 3.0).reshape(-1)),
                         "material": material()})
        elif kind == 1:                                        # a quad or a tetrahedron: one leaf
            if rng.random() < 0.5:
                p, n, i = sc.flat_mesh(sc._SQUARE_POS, sc._SQUARE_TRIS)
            else:
                p, n, i = sc.flat_mesh([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], [[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]])
            objs.append({"kind": "mesh", "pos": p, "nrm": n, "idx": i, "T": sc._colmajor(pose(rng.uniform(0.15, 0.4), t, 3.0).reshape(-1)),
                         "material": material(), "is_light": False})
        else:                                                  # blob: 8, 32 or 128 triangles
            v, f = sc.blob_mesh(int(rng.integers(0, 3)), seed=int(rng.integers(1, 1000)), radius=float(rng.uniform(0.08, 0.2)))
            p, n, i = sc.flat_mesh(v, f)
            objs.append({"kind": "mesh", "pos": p, "nrm": n, "idx": i, "T": sc._colmajor(pose(rng.uniform(0.7, 1.3), t, 3.0).reshape(-1)),
                         "material": material(), "is_light": False})
    light = dict(base["objects"][7])
    mats.append(sc._mat(sc.DIFFUSE_LIGHT, rng.uniform(4.0, 12.0, 3)))
    light["material"] = len(mats) - 1
    T = np.asarray(light["T"], np.float64).reshape(4, 4).T @ pose(rng.uniform(0.8, 1.4), np.zeros(3), 0.05)
    light["T"] = sc._colmajor(T.reshape(-1))
    objs.insert(int(rng.integers(0, len(objs) + 1)), light)
    scene = {"name": f"random{seed}", "materials": mats, "objects": objs, "camera": dict(base["camera"])}
    w, h = int(rng.integers(20, 44)), int(rng.integers(16, 36))
    out = (scene, w, h, int(rng.integers(1, 9)), bool(rng.random() < 0.85), int(rng.integers(2, 7)))
    # seeds >= 100000 also draw delta lights (point / spot / directional, Pathtracer::point_lighting) and uniform environment
    # lights; drawn last so that the scenes of the smaller seeds stay what they were
    if seed >= 100000:
        lights = []
        for _ in range(int(rng.integers(0, 6)) if rng.random() < 0.6 else 0):
            kind = int(rng.integers(0, 3))
            L = {"type": kind, "radiance": rng.uniform(0.1, 2.5, 3).astype(np.float32),
                 "T": sc._colmajor(pose(1.0, rng.uniform([-0.4, 0.2, -0.4], [0.4, 0.95, 0.4]), 0.6).reshape(-1))}
            if kind == 2:
                a = float(rng.uniform(10.0, 60.0))
                L["angle_bounds"] = np.array([a, a + float(rng.uniform(5.0, 50.0))], np.float32)
            lights.append(L)
        if lights:
            scene["lights"] = lights
        if rng.random() < 0.3:
            scene["env"] = {"type": int(rng.integers(1, 3)), "radiance": rng.uniform(0.1, 1.2, 3).astype(np.float32)}
        elif seed >= 200000 and rng.random() < 0.5:          # image environment map (Env_Map)
            eh, ew = int(rng.integers(2, 24)), int(rng.integers(2, 40))
            scene["env"] = {"type": 3, "image": np.ascontiguousarray(rng.uniform(0.0, 1.5, (eh, ew, 3)), np.float32)}
    return out
