// TEST-ONLY: the device traversal headers compiled for the host (one lane at a time) behind a tiny C API, so the
// flattened walk (pt_flat.h) can be compared with the nested form (scene_hit, pt_trace.h) and with the oracle on
// the CPU.  Built by tests/_harness.py with g++ -ffp-contract=off; uses the product's own scene builder.
#include "pt_flat.h"
#include "pt_scene.h"

#include <vector>

using namespace srt;

struct Emu {
  std::vector<ObjectInput> inputs;
  std::vector<Material> materials;
  BuiltScene built;
  DScene S;
};

extern "C" {

void* emu_create() { return new Emu(); }
void emu_destroy(void* h) { delete (Emu*)h; }

int emu_add_material(void* h, uint32_t type, const float* a, const float* b, float ior) {
  Emu* e = (Emu*)h;
  Material m; m.type = type; m.ior = ior;
  for (int i = 0; i < 3; i++) { m.a[i] = a[i]; m.b[i] = b[i]; }
  e->materials.push_back(m);
  return (int)e->materials.size() - 1;
}
int emu_add_mesh(void* h, const float* pos, const float* nrm, uint32_t nv, const uint32_t* idx, uint32_t ni, const float* T,
                 uint32_t material, int is_light) {
  Emu* e = (Emu*)h;
  ObjectInput o; o.kind = OBJ_MESH; o.material = material; o.is_light = is_light != 0;
  for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) o.trans.c[c][r] = T[4 * c + r];
  o.mesh.pos.assign(pos, pos + 3 * nv); o.mesh.nrm.assign(nrm, nrm + 3 * nv); o.mesh.idx.assign(idx, idx + ni);
  e->inputs.push_back(o);
  return 0;
}
int emu_add_sphere(void* h, float radius, const float* T, uint32_t material) {
  Emu* e = (Emu*)h;
  ObjectInput o; o.kind = OBJ_SPHERE; o.material = material; o.radius = radius;
  for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) o.trans.c[c][r] = T[4 * c + r];
  e->inputs.push_back(o);
  return 0;
}
int emu_commit(void* h, int use_bvh) {
  Emu* e = (Emu*)h;
  if (!build_scene(e->inputs, e->materials, use_bvh != 0, &e->built).empty()) return -1;
  const FlatScene& F = e->built.flat;
  DScene& S = e->S;
  S.nodes = F.nodes.data(); S.tris = F.tris.data(); S.tri_nrm = F.tri_nrm.data(); S.objects = F.objects.data();
  S.lights = F.lights.data(); S.light_tris = F.light_tris.data(); S.materials = F.materials.data();
  S.wave_tlas = F.wave_tlas.data(); S.blas_recs = F.blas_recs.data(); S.wave_lazy = F.wave_lazy.data();
  S.wave_q = (uint32_t)F.wave_tlas.size();
  S.nobjects = (uint32_t)F.objects.size(); S.nlights = (uint32_t)F.lights.size(); S.tlas_nodes = F.tlas_nodes;
  S.use_bvh = F.use_bvh ? 1u : 0u; S.light_tri_first = F.light_tri_first;
  S.delta_lights = nullptr; S.ndelta = 0;
  S.env_type = 0; S.env_map = nullptr; S.env_w = S.env_h = 0;
  S.w = S.h = 1; S.max_depth = 8;
  return 0;
}

// out: 4 values per ray {hit, dist bits, obj, tri}, for the nested form and for the flattened walk
// (each ray is traced alone in slot `slot` of a batch, the other two slots inactive).
int emu_hit(void* h, const float* org, const float* dir, const float* bounds, size_t n, int slot, uint32_t* nested, uint32_t* flat) {
  Emu* e = (Emu*)h;
  for (size_t i = 0; i < n; i++) {
    Ray r; r.o = v3p(org + 3 * i); r.d = v3p(dir + 3 * i); r.b0 = bounds[2 * i]; r.b1 = bounds[2 * i + 1];
    Counters cnt;
    for (int k = 0; k < C_COUNT; k++) cnt.v[k] = 0;
    const Hit a = scene_hit<false>(e->S, r, cnt);
    nested[4 * i] = a.hit; nested[4 * i + 1] = __float_as_uint(a.dist); nested[4 * i + 2] = a.obj; nested[4 * i + 3] = a.tri;
    Hit res[3];
    const V3 z = v3(0, 0, 1);
    flat_trace3(e->S, r.o, slot == 0 ? r.d : z, slot == 1 ? r.d : z, slot == 2 ? r.d : z, r.b0, r.b1, slot == 0, slot == 1,
                slot == 2, res[0], res[1], res[2]);
    const Hit b = res[slot];
    flat[4 * i] = b.hit; flat[4 * i + 1] = __float_as_uint(b.dist); flat[4 * i + 2] = b.obj; flat[4 * i + 3] = b.tri;
  }
  return 0;
}

// All three slots at once (shared origin): flat[12 per batch].
int emu_hit3(void* h, const float* org, const float* dir3, const float* bounds, size_t n, uint32_t* flat) {
  Emu* e = (Emu*)h;
  for (size_t i = 0; i < n; i++) {
    Hit res[3];
    flat_trace3(e->S, v3p(org + 3 * i), v3p(dir3 + 9 * i), v3p(dir3 + 9 * i + 3), v3p(dir3 + 9 * i + 6), bounds[2 * i],
                bounds[2 * i + 1], true, true, true, res[0], res[1], res[2]);
    for (int s = 0; s < 3; s++) {
      uint32_t* o = flat + 12 * i + 4 * s;
      o[0] = res[s].hit; o[1] = __float_as_uint(res[s].dist); o[2] = res[s].obj; o[3] = res[s].tri;
    }
  }
  return 0;
}

}  // extern "C"
