"""Synthetic scene inputs for the path-tracer hot path (tests, smoke, bench).

The Cornell box below is the geometry of the reference's ``media/cbox.dae`` /
``media/cbox_lambertian.dae`` (/root/reference/Assignments/Scotty3D/media/), typed in from the
COLLADA text: five unit squares placed by node matrices, a 0.5 x 0.5 area light at y = 0.999 with
radiance 10, and two analytic spheres of radius 0.2 (material tag SPHERESHAPE, scene.cpp:435-439).
Meshes are flat shaded the way ``Halfedge_Mesh::to_mesh(split_faces)`` emits them: three unique
vertices per triangle carrying the face normal (geometry/halfedge.cpp:285-297).

Scenes are plain dicts of numpy arrays — the same description is fed to the HIP path (C ABI),
to the CPU oracle and to the reference build, so all three see bit-identical inputs.

Matrices are column-major 16-float arrays (``Mat4::data`` order, lib/mat4.h).
"""
from __future__ import annotations

import numpy as np

LAMBERTIAN, MIRROR, GLASS, DIFFUSE_LIGHT, REFRACT = 0, 1, 2, 3, 4

F = np.float32


def _colmajor(rows) -> np.ndarray:
    """COLLADA stores matrices row-major; Mat4 is column-major."""
    return np.asarray(rows, F).reshape(4, 4).T.copy().reshape(16)


def to_linear(c) -> np.ndarray:
    """Spectrum::to_linear (lib/spectrum.h:56-62), applied by build_scene to Lambertian albedos."""
    c = np.asarray(c, F)
    hi = np.power((c + F(0.055)) / F(1.055), F(2.4), dtype=F)
    lo = c / F(12.92)
    return np.where(c > F(0.04045), hi, lo).astype(F)


def flat_mesh(positions, triangles):
    """positions (n,3), triangles (m,3) -> (pos (3m,3), nrm (3m,3), idx (3m,)) with face normals
    n = cross(v1 - v0, v2 - v0).unit() evaluated in float32 like lib/vec3.h."""
    P = np.asarray(positions, F)
    pos, nrm = [], []
    for t in np.asarray(triangles, np.int64):
        v0, v1, v2 = P[t[0]], P[t[1]], P[t[2]]
        a, b = (v1 - v0).astype(F), (v2 - v0).astype(F)
        c = np.array(
            [F(a[1] * b[2]) - F(a[2] * b[1]), F(a[2] * b[0]) - F(a[0] * b[2]), F(a[0] * b[1]) - F(a[1] * b[0])], F
        )
        n2 = F(F(F(c[0] * c[0]) + F(c[1] * c[1])) + F(c[2] * c[2]))
        ln = np.sqrt(n2, dtype=F)
        n = (c / ln).astype(F)
        pos += [v0, v1, v2]
        nrm += [n, n, n]
    pos = np.asarray(pos, F).reshape(-1, 3)
    nrm = np.asarray(nrm, F).reshape(-1, 3)
    return pos, nrm, np.arange(len(pos), dtype=np.uint32)


_SQUARE_POS = [[-0.5, 0, -0.5], [-0.5, 0, 0.5], [0.5, 0, -0.5], [0.5, 0, 0.5]]
_SQUARE_TRIS = [[2, 0, 1], [3, 2, 1]]
_LIGHT_POS = [[0.25, 0, -0.25], [-0.25, 0, -0.25], [-0.25, 0, 0.25], [0.25, 0, 0.25]]
_LIGHT_TRIS = [[2, 0, 1], [2, 3, 0]]

# node matrices of media/cbox.dae (row-major as printed in the file)
_M_LEFT = [1.1924876e-08, 0.99999964, 0, -0.5, -0.99999964, 1.1924876e-08, 0, 0.5, 0, 0, 1, 0, 0, 0, 0, 1]
_M_RIGHT = [-4.3711374e-08, -0.99999964, 0, 0.5, 0.99999964, -4.3711374e-08, 0, 0.5, 0, 0, 0.99999994, 0, 0, 0, 0, 1]
_M_CEIL = [-0.99999976, 1.5179339e-06, 0, 0, -1.5179339e-06, -0.99999976, 0, 1, 0, 0, 1, 0, 0, 0, 0, 1]
_M_BACK = [1, 0, 0, 0, 0, 3.1391647e-07, -1, 0.5, 0, 1, 3.1391647e-07, -0.5, 0, 0, 0, 1]
_M_FLOOR = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
_M_SPH1 = [1, 0, 0, -0.16451439, 0, 1, 0, 0.19999993, 0, 0, 1, -0.026400745, 0, 0, 0, 1]
_M_SPH2 = [1, 0, 0, 0.21465528, 0, 1, 0, 0.19999993, 0, 0, 1, 0.28486866, 0, 0, 0, 1]
_M_LIGHT = [1, 0, 0, 0, 0, 1, 0, 0.99900001, 0, 0, 1, 0, 0, 0, 0, 1]
# S3D-RENDER_CAM_NODE: used directly as the camera's iview (the GUI rebuilds it from look_at)
_M_CAM = [0.99990255, -0.0016048584, 0.013870374, -0.0034877514, 0, 0.9933728, 0.11493724, 0.61892176,
          -0.013962911, -0.11492603, 0.99327594, 1.1244006, 0, 0, 0, 1]


def _mat(kind, a=(0, 0, 0), b=(0, 0, 0), ior=1.0):
    return {"type": kind, "a": np.asarray(a, F), "b": np.asarray(b, F), "ior": float(ior)}


def _mesh(pos, tris, T, material, is_light=False):
    p, n, i = flat_mesh(pos, tris)
    return {"kind": "mesh", "pos": p, "nrm": n, "idx": i, "T": _colmajor(T), "material": material, "is_light": is_light}


def _sphere(radius, T, material):
    return {"kind": "sphere", "radius": float(radius), "T": _colmajor(T), "material": material}


def cornell_box(variant: str = "cbox", vfov: float = 90.0, ar: float = 1.0) -> dict:
    """variant 'cbox' (mirror + glass spheres) or 'cbox_lambertian' (both spheres white Lambertian).

    Camera: vfov 90 degrees (the file's xfov 121.285 at AR 1.7778, gui/render.cpp:66-71) and ar = 1
    for the square images of BASELINE configs 3-5 ("Set AR via W/H", gui/widgets.cpp:662-664)."""
    assert variant in ("cbox", "cbox_lambertian")
    white = to_linear([1, 1, 1])
    mats = [
        _mat(LAMBERTIAN, to_linear([0.7474227, 0.26583591, 0.26583591])),  # 0 left, red
        _mat(LAMBERTIAN, to_linear([0.29386759, 0.29386783, 0.81443298])),  # 1 right, blue
        _mat(LAMBERTIAN, white),  # 2 ceiling
        _mat(LAMBERTIAN, white),  # 3 back
        _mat(LAMBERTIAN, white),  # 4 floor
    ]
    if variant == "cbox":
        mats += [_mat(MIRROR, [1, 1, 1]), _mat(GLASS, [1, 1, 1], [1, 1, 1], 1.5)]
    else:
        mats += [_mat(LAMBERTIAN, white), _mat(LAMBERTIAN, white)]
    mats += [_mat(DIFFUSE_LIGHT, [10, 10, 10])]  # 7: Material::emissive() = (10/10) * 10
    objs = [
        _mesh(_SQUARE_POS, _SQUARE_TRIS, _M_LEFT, 0),
        _mesh(_SQUARE_POS, _SQUARE_TRIS, _M_RIGHT, 1),
        _mesh(_SQUARE_POS, _SQUARE_TRIS, _M_CEIL, 2),
        _mesh(_SQUARE_POS, _SQUARE_TRIS, _M_BACK, 3),
        _mesh(_SQUARE_POS, _SQUARE_TRIS, _M_FLOOR, 4),
        _sphere(0.2, _M_SPH1, 5),
        _sphere(0.2, _M_SPH2, 6),
        _mesh(_LIGHT_POS, _LIGHT_TRIS, _M_LIGHT, 7, is_light=True),
    ]
    cam = {"iview": _colmajor(_M_CAM), "vfov": float(vfov), "ar": float(ar)}
    return {"name": variant, "materials": mats, "objects": objs, "camera": cam}


def blob_mesh(n_subdiv: int, seed: int = 7, radius: float = 0.22):
    """Seeded procedural closed mesh: an octahedron subdivided n_subdiv times (8 * 4^n triangles) and
    displaced radially by a few fixed low-frequency lobes.  Stand-in for the Stanford dragon of
    BASELINE configs[4], whose .dae is a missing large blob in the reference checkout."""
    rng = np.random.default_rng(seed)
    v = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]
    f = [[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]]
    v = np.asarray(v, np.float64)
    f = np.asarray(f, np.int64)
    for _ in range(n_subdiv):
        edges = {}
        nv = [v]
        cnt = len(v)
        newf = []

        def mid(a, b):
            nonlocal cnt
            k = (min(a, b), max(a, b))
            if k not in edges:
                m = v[a] + v[b]
                nv.append((m / np.linalg.norm(m))[None])
                edges[k] = cnt
                cnt += 1
            return edges[k]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            newf += [[a, ab, ca], [ab, b, bc], [ca, bc, c], [ab, bc, ca]]
        v = np.concatenate(nv)
        f = np.asarray(newf, np.int64)
    lobes = rng.normal(size=(6, 3))
    lobes /= np.linalg.norm(lobes, axis=1, keepdims=True)
    amp = rng.uniform(0.05, 0.18, size=6)
    freq = rng.integers(2, 6, size=6)
    r = 1.0 + sum(a * np.cos(k * np.arccos(np.clip(v @ l, -1, 1))) for a, k, l in zip(amp, freq, lobes))
    v = v * (radius * r)[:, None]
    return v.astype(F), f


def cornell_with_mesh(n_subdiv: int = 7, material: str = "glass") -> dict:
    """BASELINE configs[4] stand-in: Cornell walls + light + mirror sphere + one large triangle mesh
    (8 * 4^n_subdiv triangles; n_subdiv = 7 -> 131 072) with a glass or mirror BSDF."""
    s = cornell_box("cbox")
    pos, tris = blob_mesh(n_subdiv)
    p, n, i = flat_mesh(pos, tris) if len(tris) <= 4096 else _flat_mesh_fast(pos, tris)
    T = [1, 0, 0, 0.12, 0, 1, 0, 0.30, 0, 0, 1, 0.20, 0, 0, 0, 1]
    mat = 6 if material == "glass" else 5
    s["objects"][6] = {"kind": "mesh", "pos": p, "nrm": n, "idx": i, "T": _colmajor(T), "material": mat, "is_light": False}
    s["name"] = f"cbox+blob{len(tris)}"
    return s


def cornell_with_asset(positions, triangles, T_rows, material: str = "glass", name: str = "asset") -> dict:
    """The Cornell box with an indexed triangle mesh (positions N x 3, triangles M x 3) in place of the glass sphere, posed by
    the row-major 4 x 4 `T_rows` (rotation / scale / translation: Object::hit's transform path), flat shaded."""
    s = cornell_box("cbox")
    p, n, i = _flat_mesh_fast(np.asarray(positions, F), np.asarray(triangles))
    mat = 6 if material == "glass" else 5
    s["objects"][6] = {"kind": "mesh", "pos": p, "nrm": n, "idx": i, "T": _colmajor(T_rows), "material": mat, "is_light": False}
    s["name"] = f"cbox+{name}{len(triangles)}"
    return s


def _flat_mesh_fast(positions, triangles):
    """Vectorised flat_mesh (same float32 operation order)."""
    P = np.asarray(positions, F)
    t = np.asarray(triangles, np.int64)
    v0, v1, v2 = P[t[:, 0]], P[t[:, 1]], P[t[:, 2]]
    a, b = (v1 - v0).astype(F), (v2 - v0).astype(F)
    c = np.stack(
        [
            (a[:, 1] * b[:, 2]).astype(F) - (a[:, 2] * b[:, 1]).astype(F),
            (a[:, 2] * b[:, 0]).astype(F) - (a[:, 0] * b[:, 2]).astype(F),
            (a[:, 0] * b[:, 1]).astype(F) - (a[:, 1] * b[:, 0]).astype(F),
        ],
        axis=1,
    ).astype(F)
    n2 = ((c[:, 0] * c[:, 0]).astype(F) + (c[:, 1] * c[:, 1]).astype(F)).astype(F) + (c[:, 2] * c[:, 2]).astype(F)
    n = (c / np.sqrt(n2.astype(F), dtype=F)[:, None]).astype(F)
    pos = np.stack([v0, v1, v2], axis=1).reshape(-1, 3).astype(F)
    nrm = np.repeat(n, 3, axis=0).astype(F)
    return pos, nrm, np.arange(len(pos), dtype=np.uint32)
