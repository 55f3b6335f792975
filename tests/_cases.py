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
