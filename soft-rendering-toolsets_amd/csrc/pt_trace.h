// Per-lane ("scalar") form of the path tracer's device code: closest-hit traversal with an explicit
// recursion stack, Object::hit, BSDFs, light sampling and the bounce loop of trace_pixel.  Used directly by
// the general kernels (any scene) and, piecewise, by the wave-uniform kernel in pt_wave.h.
// Reference paths are relative to /root/reference/Assignments/Scotty3D/src/.
#ifndef SRT_PT_TRACE_H
#define SRT_PT_TRACE_H

#include "pt_device.h"
#include "pt_scene.h"

namespace srt {

constexpr int kMaxTlasDepth = 24;   // interior-node nesting the traversal stacks can hold
constexpr int kMaxBlasDepth = 48;
constexpr int kMaxPathDepth = 16;   // max_depth supported by the per-bounce record stack

struct DScene {
  const Node* nodes;
  const Tri* tris;
  const TriNrm* tri_nrm;
  const Object* objects;
  const Light* lights;
  const LightTri* light_tris;
  const Material* materials;
  const WaveInterior* wave_tlas;   // interior nodes of the top-level tree in sweep order (pt_wave.h)
  const WaveInterior* blas_recs;   // interior records of every BVH<Triangle>
  const uint32_t* wave_lazy;       // per wave_tlas record: which children hold a mesh with a real BVH<Triangle> (pt_scene.h)
  const DeltaLight* delta_lights;  // Pathtracer::point_lights
  uint32_t ndelta;
  uint32_t env_type;               // Pathtracer::env_light: 0 none, 1 Env_Sphere, 2 Env_Hemisphere (uniform radiance), 3 Env_Map
  float env_radiance[3];
  const float* env_map;            // Env_Map: HDR_Image pixels, 3 floats per pixel, index y * w + x
  uint32_t env_w, env_h;
  uint32_t wave_q;
  uint32_t nobjects, nlights, tlas_nodes, use_bvh, light_tri_first;
  Camera cam;
  uint32_t w, h, max_depth;
  uint32_t elide;                  // srt_pt_set_elision and the proof holds for this scene: the dead direct ray is not traced
  // Pathtracer::log_ray (rays/pathtracer.cpp:191-193): ring of the rays the 0.0005 coin of sample_direct_lighting selects
  // (student/pathtracer.cpp:148); word 0 = rays logged so far (may exceed the capacity: the surplus is dropped), entries of
  // kRayLogWords words from word kRayLogHeader on.  NULL: nothing is logged (the coin is drawn all the same).
  uint32_t* ray_log;
  uint32_t ray_log_cap;
};

constexpr uint32_t kRayLogHeader = 8, kRayLogWords = 8;
// One logged ray: {point, dir (normalised, as Ray's constructor leaves it), pixel = y * w + x, sample << 4 | bounce}.  `key` is the
// RNG stream constant of the sample, ((pixel << 32 | sample) << 1) | 1 (Rng::key): every kernel form carries it.
SRT_DEV void log_ray_event(uint32_t* ring, uint32_t cap, float px, float py, float pz, float dx, float dy, float dz, uint64_t key, uint32_t bounce) {
  const uint32_t i = atomicAdd(ring, 1u);
  if (i < cap) {
    uint32_t* e = ring + kRayLogHeader + (size_t)kRayLogWords * i;
    e[0] = __float_as_uint(px); e[1] = __float_as_uint(py); e[2] = __float_as_uint(pz);
    e[3] = __float_as_uint(dx); e[4] = __float_as_uint(dy); e[5] = __float_as_uint(dz);
    e[6] = (uint32_t)(key >> 33);
    e[7] = ((uint32_t)(key >> 1) << 4) | (bounce & 15u);
  }
}

// Image-tile shard of one rank: tiles t with t % world == rank, numbered row-major.
struct TileMap { uint32_t tile_w, tile_h, tiles_x, tiles_y, rank, world, local_tiles; };

enum { C_RAYS = 0, C_BOX, C_OBJ, C_TRI, C_SPH, C_TLAS, C_BLAS, C_LTRI, C_COUNT };
struct Counters { uint32_t v[C_COUNT]; uint32_t elided = 0; };   // elided: rays counted in v[C_RAYS] but not traced

// Result of a closest-hit query, as ids (the payload of the winner is recomputed on demand).
struct Hit { bool hit; float dist; uint32_t obj, tri; };

// One frame of the explicit recursion stack of find_closest_hit.
struct StackFrame {
  uint32_t second;   // node to visit after the nearer child
  float fx, fy;      // cur_far_t
  uint32_t flags;    // bit0 hitboth, bit1 "nearer child done, result stored"
  float ret_dist;    // result of the nearer child (Trace::distance)
  uint32_t ret_hit, ret_a, ret_b;  // hit flag + ids
};

// Trace::min (rays/trace.h:15-23): on equal distance (or NaN) the RIGHT operand wins.
SRT_DEV bool left_wins(bool lhit, float ldist, bool rhit, float rdist) {
  if (lhit && rhit) return ldist < rdist;
  return lhit;  // only l hit -> l; only r or none -> r (none == default Trace either way)
}

// ---------------------------------------------------------------------------------------------------
// find_closest_hit as an explicit stack machine.  LeafFn(prim_slot, best) folds one primitive into
// `best` with Trace::min semantics (later wins ties).  The far child is visited iff
// cur_far_t.x < ret.distance || (!ret.hit && hitboth) where ret is the NEARER CHILD's result only —
// hits found in other subtrees never prune (student/bvh.inl:215-219); that is kept as is.
// ---------------------------------------------------------------------------------------------------
template <int MAXD, bool COUNT, typename LeafFn>
SRT_DEV Hit traverse(const Node* __restrict__ nodes, const Ray& ray, float tx, float ty, Counters& cnt, int cnt_slot,
                     LeafFn&& leaf) {
  StackFrame stack[MAXD];
  int sp = 0;
  uint32_t cur = 0;
  Hit ret;
  for (;;) {
    // ---- descend into `cur` with times (tx, ty) ----
    if (COUNT) cnt.v[cnt_slot]++;
    const Node nd = nodes[cur];
    ret.hit = false; ret.dist = 0.0f; ret.obj = 0; ret.tri = 0;
    bool descended = false;
    if (nd.count & LEAF_BIT) {
      const uint32_t n = nd.count & ~LEAF_BIT;
      for (uint32_t i = 0; i < n; i++) leaf(nd.left + i, ret);
    } else {
      float t1x = tx, t1y = ty, t2x = tx, t2y = ty;
      const Node nl = nodes[nd.left], nr = nodes[nd.left + 1];
      if (COUNT) cnt.v[C_BOX] += 2;
      const bool hl = box_hit(nl, ray, t1x, t1y);
      const bool hr = box_hit(nr, ray, t2x, t2y);
      if (hl || hr) {
        uint32_t closer, second;
        bool hitboth = false;
        float cx = ray.b0, cy = ray.b1, fx = ray.b0, fy = ray.b1;
        if (hl && hr) {
          hitboth = true;
          if (t1x < t2x) { closer = nd.left; second = nd.left + 1; cx = t1x; cy = t1y; fx = t2x; fy = t2y; }
          else { closer = nd.left + 1; second = nd.left; cx = t2x; cy = t2y; fx = t1x; fy = t1y; }
        } else if (hl) { closer = nd.left; second = nd.left + 1; cx = t1x; cy = t1y; }
        else { closer = nd.left + 1; second = nd.left; cx = t2x; cy = t2y; }
        StackFrame& f = stack[sp++];
        f.second = second; f.fx = fx; f.fy = fy; f.flags = hitboth ? 1u : 0u;
        cur = closer; tx = cx; ty = cy;
        descended = true;
      }
    }
    if (descended) continue;
    // ---- ascend: `ret` is the result of the subtree just finished ----
    bool resume = false;
    while (sp > 0) {
      StackFrame& f = stack[sp - 1];
      if (!(f.flags & 2u)) {
        // back from the nearer child
        if (f.fx < ret.dist || (!ret.hit && (f.flags & 1u))) {
          f.flags |= 2u;
          f.ret_hit = ret.hit ? 1u : 0u; f.ret_dist = ret.dist; f.ret_a = ret.obj; f.ret_b = ret.tri;
          cur = f.second; tx = f.fx; ty = f.fy;
          resume = true;
          break;
        }
        sp--;
      } else {
        // back from the farther child: ret = Trace::min(first, second)
        if (left_wins(f.ret_hit != 0, f.ret_dist, ret.hit, ret.dist)) {
          ret.hit = true; ret.dist = f.ret_dist; ret.obj = f.ret_a; ret.tri = f.ret_b;
        } else if (!ret.hit) {
          ret.hit = false; ret.dist = 0.0f; ret.obj = 0; ret.tri = 0;  // `return {}`
        }
        sp--;
      }
    }
    if (!resume) return ret;
  }
}

// Fold one candidate into `best` the way `ret = Trace::min(ret, hit)` does.
SRT_DEV void fold(Hit& best, bool hit, float dist, uint32_t obj, uint32_t tri) {
  if (left_wins(best.hit, best.dist, hit, dist)) return;
  if (hit) { best.hit = true; best.dist = dist; best.obj = obj; best.tri = tri; }
  else { best.hit = false; best.dist = 0.0f; best.obj = 0; best.tri = 0; }
}

// BBox::hit on a record's child box with the reciprocal direction hoisted, straight-line form.
SRT_DEV bool box_hit_rec(const float* __restrict__ bx, V3 o, V3 inv, float& tx, float& ty) {
  const bool sx = inv.x < 0, sy = inv.y < 0, sz = inv.z < 0;
  // the six bounds are read into values first: `c ? bx[3] : bx[0]` on the memory operands is an lvalue select,
  // i.e. one per-lane vector load from a selected address instead of two wave-uniform scalar loads
  const float b0 = bx[0], b1 = bx[1], b2 = bx[2], b3 = bx[3], b4 = bx[4], b5 = bx[5];
  float tmin = ((sx ? b3 : b0) - o.x) * inv.x;
  float tmax = ((sx ? b0 : b3) - o.x) * inv.x;
  const float tymin = ((sy ? b4 : b1) - o.y) * inv.y;
  const float tymax = ((sy ? b1 : b4) - o.y) * inv.y;
  const bool miss_y = (tmin > tymax) || (tymin > tmax);
  tmin = (tymin > tmin) ? tymin : tmin;
  tmax = (tymax < tmax) ? tymax : tmax;
  const float tzmin = ((sz ? b5 : b2) - o.z) * inv.z;
  const float tzmax = ((sz ? b2 : b5) - o.z) * inv.z;
  const bool miss_z = (tmin > tzmax) || (tzmin > tmax);
  tmin = (tzmin > tmin) ? tzmin : tmin;
  tmax = (tzmax < tmax) ? tzmax : tmax;
  const bool hit = !miss_y && !miss_z;
  const float nx = (tmin >= tx && tmin <= ty) ? tmin : tx;
  const float ny = (tmax >= nx && tmax <= ty) ? tmax : ty;
  tx = hit ? nx : tx;
  ty = hit ? ny : ty;
  return hit;
}

// find_closest_hit of one BVH<Triangle> over its interior records: one 64-byte fetch brings both child boxes,
// leaves are folded in place, a stack frame is 16 bytes (after the nearer child returns, its result reuses
// the slots of cur_far_t).  Same visit rule and Trace::min as traverse<>().
struct RecFrame { int32_t second; float a, b; uint32_t flags; };  // flags: 1 hitboth, 2 nearer child done, 4 its hit
template <int MAXD, bool COUNT>
SRT_DEV Hit traverse_records(const DScene& S, const Object& o, const Ray& ray, float tx, float ty, Counters& cnt) {
  const WaveInterior* __restrict__ recs = S.blas_recs + o.rec_base;
  const V3 inv = v3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
  RecFrame stack[MAXD];
  int sp = 0;
  int32_t cur = 0;
  Hit ret;
  for (;;) {
    if (COUNT) cnt.v[C_BLAS]++;
    ret.hit = false; ret.dist = 0.0f; ret.obj = 0; ret.tri = 0;
    bool descended = false;
    if (cur < 0) {                                   // leaf: fold its triangles in order
      const uint32_t packed = (uint32_t)~cur;
      const uint32_t first = o.tri_base + (packed >> 3), n = packed & 7u;
      for (uint32_t i = 0; i < n; i++) {
        if (COUNT) cnt.v[C_TRI]++;
        const TriHit th = tri_hit(S.tris[first + i], ray);
        fold(ret, th.hit, th.dist, 0, first + i);
      }
    } else {
      const WaveInterior W = recs[cur];
      if (COUNT) cnt.v[C_BOX] += 2;
      float t1x = tx, t1y = ty, t2x = tx, t2y = ty;
      const bool hl = box_hit_rec(W.boxl, ray.o, inv, t1x, t1y);
      const bool hr = box_hit_rec(W.boxr, ray.o, inv, t2x, t2y);
      if (hl || hr) {
        const bool hb = hl && hr;
        const bool cl = hb ? (t1x < t2x) : hl;
        RecFrame& f = stack[sp++];
        f.second = cl ? W.r_ref : W.l_ref;
        f.a = hb ? (cl ? t2x : t1x) : ray.b0;
        f.b = hb ? (cl ? t2y : t1y) : ray.b1;
        f.flags = hb ? 1u : 0u;
        cur = cl ? W.l_ref : W.r_ref;
        tx = cl ? t1x : t2x;
        ty = cl ? t1y : t2y;
        descended = true;
      }
    }
    if (descended) continue;
    bool resume = false;
    while (sp > 0) {
      RecFrame& f = stack[sp - 1];
      if (!(f.flags & 2u)) {
        if (f.a < ret.dist || (!ret.hit && (f.flags & 1u))) {
          cur = f.second; tx = f.a; ty = f.b;
          f.flags |= 2u | (ret.hit ? 4u : 0u);
          f.a = ret.dist; f.b = __uint_as_float(ret.tri);
          resume = true;
          break;
        }
        sp--;
      } else {
        if (left_wins((f.flags & 4u) != 0, f.a, ret.hit, ret.dist)) {
          ret.hit = true; ret.dist = f.a; ret.obj = 0; ret.tri = __float_as_uint(f.b);
        } else if (!ret.hit) {
          ret.hit = false; ret.dist = 0.0f; ret.obj = 0; ret.tri = 0;
        }
        sp--;
      }
    }
    if (!resume) return ret;
  }
}

// The same walk for lanes that walk together (the compacted rounds of the wave kernel, pt_wave.h): "while-while" form.
// The wave repeats interior steps while any lane stands at an interior record - lanes that have reached a leaf wait -,
// then every lane that is still walking tests the triangles of its leaf; a single loop with `leaf ? tests : boxes`
// would pay for both bodies on every step with about half of the lanes in each.  Same records, frames, visit rule and
// results as traverse_records.
template <int MAXD>
SRT_DEV Hit traverse_records_together(const DScene& S, const Object& o, const Ray& ray, float tx, float ty) {
  const WaveInterior* __restrict__ recs = S.blas_recs + o.rec_base;
  const V3 inv = v3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
  RecFrame stack[MAXD];
  int sp = 0;
  int32_t cur = 0;
  bool done = false;
  Hit ret;
  ret.hit = false; ret.dist = 0.0f; ret.obj = 0; ret.tri = 0;
  // after a node has produced `ret`: pop finished frames; stop at one whose second child has to be visited (-> cur) or when
  // the stack is empty (-> done, ret is the answer)
  auto unwind = [&]() {
    while (sp > 0) {
      RecFrame& f = stack[sp - 1];
      if (!(f.flags & 2u)) {
        if (f.a < ret.dist || (!ret.hit && (f.flags & 1u))) {
          cur = f.second; tx = f.a; ty = f.b;
          f.flags |= 2u | (ret.hit ? 4u : 0u);
          f.a = ret.dist; f.b = __uint_as_float(ret.tri);
          return;
        }
        sp--;
      } else {
        if (left_wins((f.flags & 4u) != 0, f.a, ret.hit, ret.dist)) {
          ret.hit = true; ret.dist = f.a; ret.obj = 0; ret.tri = __float_as_uint(f.b);
        } else if (!ret.hit) {
          ret.hit = false; ret.dist = 0.0f; ret.obj = 0; ret.tri = 0;
        }
        sp--;
      }
    }
    done = true;
  };
  for (;;) {
    while (__ballot(!done && cur >= 0) != 0ull) {      // interior steps, together
      if (!done && cur >= 0) {
        const WaveInterior W = recs[cur];
        float t1x = tx, t1y = ty, t2x = tx, t2y = ty;
        const bool hl = box_hit_rec(W.boxl, ray.o, inv, t1x, t1y);
        const bool hr = box_hit_rec(W.boxr, ray.o, inv, t2x, t2y);
        if (hl || hr) {
          const bool hb = hl && hr;
          const bool cl = hb ? (t1x < t2x) : hl;
          RecFrame& f = stack[sp++];
          f.second = cl ? W.r_ref : W.l_ref;
          f.a = hb ? (cl ? t2x : t1x) : ray.b0;
          f.b = hb ? (cl ? t2y : t1y) : ray.b1;
          f.flags = hb ? 1u : 0u;
          cur = cl ? W.l_ref : W.r_ref;
          tx = cl ? t1x : t2x;
          ty = cl ? t1y : t2y;
        } else {
          ret.hit = false; ret.dist = 0.0f; ret.obj = 0; ret.tri = 0;
          unwind();
        }
      }
    }
    if (__ballot(!done) == 0ull) break;
    if (!done) {                                        // every lane still walking stands at a leaf
      const uint32_t packed = (uint32_t)~cur;
      const uint32_t first = o.tri_base + (packed >> 3), n = packed & 7u;
      ret.hit = false; ret.dist = 0.0f; ret.obj = 0; ret.tri = 0;
      for (uint32_t i = 0; i < n; i++) {
        const TriHit th = tri_hit(S.tris[first + i], ray);
        fold(ret, th.hit, th.dist, 0, first + i);
      }
      unwind();
    }
  }
  return ret;
}

// Closest triangle of one mesh in OBJECT space: Tri_Mesh::hit -> BVH<Triangle>::hit / List<Triangle>::hit.
// Returns ids in Hit (tri = global triangle index).
template <bool COUNT, bool TOGETHER = false>
SRT_DEV Hit mesh_hit(const DScene& S, const Object& o, const Ray& oray, Counters& cnt) {
  Hit best; best.hit = false; best.dist = 0.0f; best.obj = 0; best.tri = 0;
  if (o.use_bvh && o.nrec > 0) {
    const float dn = norm(oray.d);
    if (TOGETHER && !COUNT) return traverse_records_together<kMaxBlasDepth>(S, o, oray, oray.b0 / dn, oray.b1 / dn);
    return traverse_records<kMaxBlasDepth, COUNT>(S, o, oray, oray.b0 / dn, oray.b1 / dn, cnt);  // times = dist_bounds / dir.norm()
  }
  if (o.use_bvh) {
    if (o.nnodes == 0) return best;
    if (COUNT) cnt.v[C_BLAS]++;                      // the root is a leaf holding every triangle
  }
  for (uint32_t t = 0; t < o.ntri; t++) {
    if (COUNT) cnt.v[C_TRI]++;
    const TriHit th = tri_hit(S.tris[o.tri_base + t], oray);
    fold(best, th.hit, th.dist, 0, o.tri_base + t);
  }
  return best;
}

// Object::hit (rays/object.h:57-65) reduced to what closest-hit selection needs: hit flag and the
// WORLD distance Trace::transform recomputes (|T*position - T*origin|, rays/trace.h:25-30).
template <bool COUNT>
SRT_DEV void object_hit(const DScene& S, uint32_t slot, const Ray& wray, Hit& acc, Counters& cnt) {
  const Object& o = S.objects[slot];
  Ray ray = wray;
  if (o.has_trans) {
    if (COUNT) cnt.v[C_OBJ]++;
    ray_transform(ray, o.itrans);
  }
  bool hit;
  float dist;
  uint32_t tri = 0;
  V3 pos;
  if (o.kind == OBJ_SPHERE) {
    if (COUNT) cnt.v[C_SPH]++;
    const SphHit sh = sphere_hit(o.radius, ray);
    hit = sh.hit;
    pos = ray_at(ray, sh.t);
    dist = fabsf(norm(pos - ray.o));
  } else {
    const Hit mh = mesh_hit<COUNT>(S, o, ray, cnt);
    hit = mh.hit;
    dist = mh.dist;
    tri = mh.tri;
    if (hit && o.has_trans) {
      const TriHit th = tri_hit(S.tris[tri], ray);  // (u,v,t) of the winner; same arithmetic, same bits
      pos = ray_at(ray, th.t);
    }
  }
  if (hit && o.has_trans) {
    const V3 pw = mat_point(o.trans, pos);
    const V3 ow = mat_point(o.trans, ray.o);
    dist = norm(pw - ow);
  }
  fold(acc, hit, dist, slot, tri);
}

// scene.hit(ray): BVH<Object>::hit or List<Object>::hit; the scene Object itself has no transform.
template <bool COUNT>
SRT_DEV Hit scene_hit(const DScene& S, const Ray& ray, Counters& cnt) {
  cnt.v[C_RAYS]++;  // always: the Mrays/s metric counts scene.hit calls
  if (S.use_bvh) {
    Hit none; none.hit = false; none.dist = 0.0f; none.obj = 0; none.tri = 0;
    if (S.tlas_nodes == 0) return none;
    const float dn = norm(ray.d);
    const float tx = ray.b0 / dn, ty = ray.b1 / dn;
    auto leaf = [&](uint32_t slot, Hit& acc) { object_hit<COUNT>(S, slot, ray, acc, cnt); };
    return traverse<kMaxTlasDepth, COUNT>(S.nodes, ray, tx, ty, cnt, C_TLAS, leaf);
  }
  Hit best; best.hit = false; best.dist = 0.0f; best.obj = 0; best.tri = 0;
  for (uint32_t i = 0; i < S.nobjects; i++) object_hit<COUNT>(S, i, ray, best, cnt);
  return best;
}

// Trace payload of the winner (position, normal) as Object::hit + Trace::transform produce it.
struct Surface { V3 position, normal; };
SRT_DEV Surface surface_of(const DScene& S, const Hit& h, const Ray& wray) {
  const Object& o = S.objects[h.obj];
  Ray ray = wray;
  if (o.has_trans) ray_transform(ray, o.itrans);
  Surface sf;
  if (o.kind == OBJ_SPHERE) {
    const SphHit sh = sphere_hit(o.radius, ray);
    sf.position = ray_at(ray, sh.t);
    sf.normal = ray_at(ray, sh.t) - v3(0.0f, 0.0f, 0.0f);
  } else {
    const TriHit th = tri_hit(S.tris[h.tri], ray);
    const TriNrm& nn = S.tri_nrm[h.tri];
    sf.position = ray_at(ray, th.t);
    // u*n0 + v*n1 + (1-u-v)*n2, the fork's own weighting (student/tri_mesh.cpp:104-106)
    sf.normal = (v3p(nn.n0) * th.u + v3p(nn.n1) * th.v) + v3p(nn.n2) * (1.0f - th.u - th.v);
  }
  if (o.has_trans) {
    sf.position = mat_point(o.trans, sf.position);
    sf.normal = unit(mat_rotate_transposed(o.itrans, sf.normal));  // itrans.T().rotate(n).unit()
  }
  return sf;
}

// ---------------------------------------------------------------------------------------------------
// BSDFs (student/bsdf.cpp) and samplers (student/samplers.cpp)
// ---------------------------------------------------------------------------------------------------
struct Scatter { Spec atten; V3 dir; };
SRT_DEV bool is_discrete(uint32_t t) { return t == 1 || t == 2 || t == 4; }
SRT_DEV bool is_sided(uint32_t t) { return t == 2 || t == 4; }
SRT_DEV V3 reflect(V3 d) { return v3((-1.0f) * d.x, d.y, (-1.0f) * d.z); }
SRT_DEV Spec lambert_evaluate(const Material& m, V3 out) {
  const V3 u = unit(out);
  const float theta = dot(u, v3(0.0f, 1.0f, 0.0f));
  return spec(m.a[0], m.a[1], m.a[2]) * srt_cosf(theta);
}
SRT_DEV float lambert_pdf(V3 out) {
  const float theta = dot(out, v3(0.0f, 1.0f, 0.0f));
  float ct = srt_cosf(theta);
  ct = std_min(std_max(ct, 0.0f), 1.0f);
  return ct / kPi;
}
// Samplers::Hemisphere::Cosine::sample (student/samplers.cpp:166-177)
SRT_DEV V3 lambert_direction(Rng& rng) {
  const float phi = rng.unit() * 2.0f * kPi;
  const float cos_t = sqrtf(rng.unit());
  const float sin_t = sqrtf(1 - cos_t * cos_t);
  float cphi, sphi;
  srt_sincosf2(phi, cphi, sphi);
  return v3(cphi * sin_t, cos_t, sphi * sin_t);
}
SRT_DEV Scatter scatter(const Material& m, V3 out, Rng& rng) {
  Scatter r;
  if (m.type == 0) {                                   // BSDF_Lambertian::scatter, bsdf.cpp:69-87
    const float phi = rng.unit() * 2.0f * kPi;         // Hemisphere::Cosine::sample, samplers.cpp:166-177
    const float cos_t = sqrtf(rng.unit());
    const float sin_t = sqrtf(1 - cos_t * cos_t);
    float cphi, sphi;
    srt_sincosf2(phi, cphi, sphi);
    const float x = cphi * sin_t;
    const float z = sphi * sin_t;
    r.dir = v3(x, cos_t, z);
    r.atten = lambert_evaluate(m, out);
  } else if (m.type == 1) {                            // BSDF_Mirror::scatter, bsdf.cpp:119-126
    r.dir = reflect(out);
    r.atten = spec(m.a[0], m.a[1], m.a[2]);
  } else if (m.type == 2) {                            // BSDF_Glass::scatter, bsdf.cpp:128-154
    const float ior = m.ior;
    const float cos_i = out.y;                         // refract(), bsdf.cpp:23-64
    float ni, nt;
    if (cos_i > 0) { nt = ior; ni = 1.0f; } else { nt = 1.0f; ni = ior; }
    const float ratio = ni / nt;
    const float cos_t_sq = 1.0f - (float)pow2d(ratio) * (1.0f - (float)pow2d(cos_i));
    const bool internal = cos_t_sq < 0;
    V3 refr;
    if (internal) {
      refr = reflect(out);
    } else {
      const float cos_t = (cos_i >= 0) ? (float)((double)(-1.0f) * sqrt((double)cos_t_sq)) : (float)sqrt((double)cos_t_sq);
      refr = v3((-1.0f) * out.x * ratio, cos_t, (-1.0f) * out.z * ratio);
    }
    float r0 = (1 - ior) / (1 + ior);                  // Schlick_Approximation, bsdf.cpp:17-21
    r0 = r0 * r0;
    const float fresnel = r0 + (1 - r0) * (float)pow5d(1 - fabsf(out.y));
    const bool flip = rng.coin(fresnel);               // always drawn (left operand of ||)
    if (flip || internal) {
      r.dir = reflect(out);
      r.atten = spec(m.b[0], m.b[1], m.b[2]);
    } else {
      r.dir = refr;
      const float rr = (out.y > 0) ? (1.0f / ior) : ior;
      r.atten = spec(m.a[0], m.a[1], m.a[2]) * (float)pow2d(rr);
    }
  } else {                                             // BSDF_Refract stub, bsdf.cpp:156-166
    r.dir = v3(0, 0, 0);
    r.atten = spec(0, 0, 0);
  }
  return r;
}
SRT_DEV Spec emissive_of(const Material& m) { return m.type == 3 ? spec(m.a[0], m.a[1], m.a[2]) : spec(0, 0, 0); }

SRT_DEV float std_clamp(float v, float lo, float hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }   // std::clamp
SRT_DEV Spec lerp_spec(float ratio, Spec a, Spec b) {      // lerpSpectrum (student/env_light.cpp:33-35): (1 - ratio) * start + ratio * ends
  const float om = 1 - ratio;
  return spec(om * a.r + ratio * b.r, om * a.g + ratio * b.g, om * a.b + ratio * b.b);
}
// HDR_Image::at(x, y) on the environment map; the float -> int conversions of the reference (x86: INT_MIN for NaN / out
// of range, an assert there) are clamped into the image here.
SRT_DEV Spec env_texel(const DScene& S, float x, float y) {
  int xi = (x >= 0.0f && x < 2147483648.0f) ? (int)x : 0, yi = (y >= 0.0f && y < 2147483648.0f) ? (int)y : 0;
  xi = xi < (int)S.env_w ? xi : (int)S.env_w - 1;
  yi = yi < (int)S.env_h ? yi : (int)S.env_h - 1;
  const float* p = S.env_map + 3 * ((size_t)yi * S.env_w + (size_t)xi);
  return spec(p[0], p[1], p[2]);
}
// Env_Map::evaluate (student/env_light.cpp:37-93): direction -> (theta, phi) -> bilinear lookup as the fork does it.
SRT_DEV Spec env_map_evaluate(const DScene& S, V3 dir) {
  const float r = norm(dir);
  float theta = kPi - srt_acosf(dir.y / r);
  float phi = srt_atan2f(dir.z, dir.x);
  if (phi < 0) phi = phi + 2.f * kPi;
  theta = std_clamp(theta / kPi, 0.f, 1.f);
  phi = std_clamp(phi / (2.0f * kPi), 0.f, 1.f);
  const float h = (float)S.env_h, w = (float)S.env_w;
  const float u = phi * w, v = theta * h;
  float u0 = floorf(u), v0 = floorf(v), u1, v1;
  if (u - u0 < 0.5f) { u1 = u0; u0 = u0 - 1; } else { u1 = u0 + 1.f; }     // u1 = u0 - 1; swap(u0, u1)
  if (v - v0 < 0.5f) { v1 = v0; v0 = v0 - 1; } else { v1 = v0 + 1.f; }
  u0 = std_min(std_max(u0, 0.f), w - 1.f); u1 = std_min(std_max(u1, 0.f), w - 1.f);   // clamp(): min(max(x, lo), hi)
  v0 = std_min(std_max(v0, 0.f), h - 1.f); v1 = std_min(std_max(v1, 0.f), h - 1.f);
  const float ru = std_min(std_max(u - u0 - 0.5f, 0.f), 1.f), rv = std_min(std_max(v - v0 - 0.5f, 0.f), 1.f);
  const Spec h1 = lerp_spec(ru, env_texel(S, u0, v0), env_texel(S, u1, v0));
  const Spec h2 = lerp_spec(ru, env_texel(S, u0, v1), env_texel(S, u1, v1));
  return lerp_spec(rv, h1, h2);
}

// Env_Sphere / Env_Hemisphere / Env_Map::evaluate (student/env_light.cpp:37-118).
SRT_DEV Spec env_evaluate(const DScene& S, V3 dir) {
  if (S.env_type == 3u) return env_map_evaluate(S, dir);
  const Spec r = spec(S.env_radiance[0], S.env_radiance[1], S.env_radiance[2]);
  if (S.env_type == 2u) return (dir.y > 0.0f) ? r : spec(0, 0, 0);
  return r;
}
// Env_*::sample = Samplers::Hemisphere::Uniform::sample (student/samplers.cpp:151-164; Sphere::Uniform returns the
// same upper-hemisphere sample, :17-26): two draws.
SRT_DEV V3 env_sample(Rng& rng) {
  const float xi1 = rng.unit();
  const float xi2 = rng.unit();
  const float theta = srt_acosf(xi1);
  const float phi = 2.0f * kPi * xi2;
  float ct, st, cp, sp;
  srt_sincosf2(theta, ct, st);
  srt_sincosf2(phi, cp, sp);
  return v3(st * cp, ct, st * sp);
}

// Pathtracer::sample_area_lights (rays/pathtracer.cpp:301-311): List<Object>::sample -> Object::sample ->
// List<Triangle>::sample -> Samplers::Triangle::sample; with an environment light a coin picks between the two.
SRT_DEV V3 light_sample(const DScene& S, V3 from, Rng& rng) {
  if (S.env_type != 0u) {
    if (S.nlights == 0) return env_sample(rng);
    if (rng.coin(0.5f)) return env_sample(rng);
  }
  if (S.nlights == 0) return v3(0, 0, 0);
  const Light& L = S.lights[rng.integer(0, (int)S.nlights)];
  if (L.has_trans) from = mat_point(L.itrans, from);
  const uint32_t t = (uint32_t)rng.integer(0, (int)L.ntri);
  const LightTri& lt = S.light_tris[L.tri_base - S.light_tri_first + t];
  const float u = sqrtf(rng.unit());
  const float v = rng.unit();
  const float a = u * (1.0f - v);
  const float b = u * v;
  const V3 pos = (v3p(lt.v0) * a + v3p(lt.v1) * b) + v3p(lt.v2) * (1.0f - a - b);
  V3 dir = unit(pos - from);
  if (L.has_trans) dir = unit(mat_rotate(L.trans, dir));
  return dir;
}
// Pathtracer::area_lights_pdf -> List<Object>::pdf -> Object::pdf -> List<Triangle>::pdf -> Triangle::pdf.
template <bool COUNT>
SRT_DEV float light_pdf(const DScene& S, V3 from, V3 dir, Counters& cnt) {
  int n = 0;
  float pdf = 0.0f;
  if (S.nlights) {
    const Ray wray = make_ray(from, dir, 0.0f, FLT_MAX);
    float ret = 0.0f;
    for (uint32_t li = 0; li < S.nlights; li++) {
      const Light& L = S.lights[li];
      float sum = 0.0f;
      Ray tray = wray;
      ray_transform(tray, L.pdfiT);             // Triangle::pdf transforms per triangle; same matrix, same result
      for (uint32_t t = 0; t < L.ntri; t++) {
        if (COUNT) cnt.v[C_LTRI]++;
        const uint32_t gi = L.tri_base + t;
        const TriHit th = tri_hit(S.tris[gi], tray);
        float p = 0.0f;
        if (th.hit) {
          const TriNrm& nn = S.tri_nrm[gi];
          V3 position = ray_at(tray, th.t);
          V3 normal = (v3p(nn.n0) * th.u + v3p(nn.n1) * th.v) + v3p(nn.n2) * (1.0f - th.u - th.v);
          position = mat_point(L.pdfT, position);                       // trace.transform(T, iT.T())
          normal = unit(mat_rotate_transposed(L.pdfiT, normal));
          const float a = S.light_tris[gi - S.light_tri_first].area_term;
          const float g = norm2(position - wray.o) / fabsf(dot(normal, wray.d));
          p = a * g;
        }
        sum += p;
      }
      ret += sum / (float)L.ntri;
    }
    pdf += ret / (float)S.nlights;
    n++;
  }
  if (S.env_type != 0u) {                          // Env_Sphere::pdf = 1 / (4 PI), Env_Hemisphere::pdf = 1 / (2 PI)
    pdf += (S.env_type == 2u) ? (1.0f / (2.0f * kPi)) : (1.0f / (4.0f * kPi));
    n++;
  }
  if (n) pdf /= n;
  return pdf;
}

// Camera::generate_ray (student/camera.cpp:7-34); screen_h/screen_w come from the host (tanf).
SRT_DEV Ray camera_ray(const Camera& cam, float sx, float sy) {
  const float sh = cam.screen_h, sw = cam.screen_w;
  Ray r;
  r.o = v3(0, 0, 0);
  r.d = v3(sx * sw - 0.5f * sw, sy * sh - 0.5f * sh, -1.0f);
  r.b0 = 0.0f;
  r.b1 = __uint_as_float(0x7f800000u);
  ray_transform(r, cam.iview);
  return r;
}
SRT_DEV Ray camera_ray(const DScene& S, float sx, float sy) { return camera_ray(S.cam, sx, sy); }

// `trace(ray).first` of a depth-0 ray: emitted radiance of whatever it hits, else zero.
template <bool COUNT>
SRT_DEV Spec emitted_along(const DScene& S, const Ray& ray, Counters& cnt) {
  const Hit h = scene_hit<COUNT>(S, ray, cnt);
  if (!h.hit) return (S.env_type != 0u) ? env_evaluate(S, ray.d) : spec(0, 0, 0);   // student/pathtracer.cpp:182-188
  const Spec e = emissive_of(S.materials[S.objects[h.obj].material]);
  return (luma(e) > 0.0f) ? e : spec(0, 0, 0);
}

// Delta_Light::sample (rays/light.h:86-91) over Directional / Point / Spot_Light::sample (rays/light.cpp:5-31).
struct LightSample { Spec radiance; V3 direction; float distance; };
SRT_DEV LightSample delta_light_sample(const DeltaLight& l, V3 from) {
  if (l.has_trans) from = mat_point(l.itrans, from);
  LightSample r;
  r.radiance = spec(l.radiance[0], l.radiance[1], l.radiance[2]);
  if (l.type == DL_DIRECTIONAL) {
    r.direction = v3(0.0f, -1.0f, 0.0f);
    r.distance = __uint_as_float(0x7f800000u);
  } else {
    r.direction = neg(unit(from));
    r.distance = norm(from);
    if (l.type == DL_SPOT) {
      float angle = srt_atan2f(sqrtf(from.x * from.x + from.z * from.z), from.y);   // atan2(Vec2(x, z).norm(), y)
      angle = fabsf(angle * (180.0f / kPi));                                          // std::abs(Degrees(angle))
      const float e0 = l.angle_bounds[0] / 2.0f, e1 = l.angle_bounds[1] / 2.0f;
      const float t = std_min(std_max((angle - e0) / (e1 - e0), 0.0f), 1.0f);         // smoothstep, lib/mathlib.h:47-50
      const float k = 1.0f - t * t * (3.0f - 2.0f * t);
      r.radiance = spec(k * r.radiance.r, k * r.radiance.g, k * r.radiance.b);       // float * Spectrum
    }
  }
  if (l.has_trans) r.direction = mat_rotate(l.trans, r.direction);
  return r;
}

// Pathtracer::point_lighting (rays/pathtracer.cpp:327-348) for a continuous BSDF: one shadow ray per delta light whose
// attenuation has luma != 0; no random numbers are drawn.
template <bool COUNT>
SRT_DEV Spec point_lighting(const DScene& S, const Material& m, V3 position, V3 out_dir, Counters& cnt) {
  Spec radiance = spec(0, 0, 0);
  for (uint32_t i = 0; i < S.ndelta; i++) {
    const LightSample ls = delta_light_sample(S.delta_lights[i], position);
    const Spec att = lambert_evaluate(m, out_dir);    // bsdf.evaluate(out_dir, in_dir): the Lambertian ignores in_dir
    if (luma(att) == 0.0f) continue;
    const Ray shadow = make_ray(position, ls.direction, kEps, ls.distance - kEps);
    const Hit h = scene_hit<COUNT>(S, shadow, cnt);
    if (!h.hit) radiance = radiance + att * ls.radiance;
  }
  return radiance;
}

struct Bounce { Spec direct, atten; float inv_pdf; uint32_t discrete; };

// Pathtracer::trace_pixel for pixel (x, y); the RNG must already be keyed.
template <bool COUNT>
SRT_DEV Spec path_sample(const DScene& S, uint32_t x, uint32_t y, Rng& rng, Counters& cnt) {
  const float jx = rng.unit() * 1.0f;   // Samplers::Rect(1,1).sample(): x first (braced init)
  const float jy = rng.unit() * 1.0f;
  Ray ray = camera_ray(S, ((float)x + jx) / (float)S.w, ((float)y + jy) / (float)S.h);
  uint32_t depth = S.max_depth;
  Spec emissive_cam = spec(0, 0, 0);
  Bounce rec[kMaxPathDepth];
  int level = 0;
  for (;;) {
    const Hit h = scene_hit<COUNT>(S, ray, cnt);
    if (!h.hit) {                                   // {env_light.evaluate(ray.dir), {}}: only a camera ray's `.first` is used
      if (level == 0 && S.env_type != 0u) emissive_cam = env_evaluate(S, ray.d);
      break;
    }
    const Material& m = S.materials[S.objects[h.obj].material];
    const Spec e = emissive_of(m);
    if (luma(e) > 0.0f) { if (level == 0) emissive_cam = e; break; }
    if (depth == 0) break;
    Surface sf = surface_of(S, h, ray);
    if (!is_sided(m.type) && dot(sf.normal, ray.d) > 0.0f) sf.normal = neg(sf.normal);
    const Frame fr = rotate_to(sf.normal);
    const V3 out_dir = unit(frame_to_local(fr, ray.o - sf.position));
    const bool discrete = is_discrete(m.type);

    // ---- sample_direct_lighting ----
    Spec radiance = spec(0, 0, 0);  // point_lighting() returns {} for a discrete BSDF
    if (!discrete && S.ndelta) radiance = point_lighting<COUNT>(S, m, sf.position, out_dir, cnt);
    const Scatter s1 = scatter(m, out_dir, rng);
    const V3 world_in = frame_to_world(fr, s1.dir);
    const Ray r1 = make_ray(sf.position, world_in, kEps, FLT_MAX);
    // S.elide (no delta / environment light, every continuous BSDF Lambertian): the term of this ray is added and subtracted
    // again below, +0 for any finite value and NaN together with the MIS term otherwise - the ray is counted, not traced
    const bool dead = S.elide != 0u && !discrete;
    Spec direct = spec(0, 0, 0);
    if (dead) { cnt.v[C_RAYS]++; cnt.elided++; }
    else direct = emitted_along<COUNT>(S, r1, cnt);
    float pdf = 0.0f;
    if (discrete) {
      direct = direct * s1.atten;
    } else {
      pdf = lambert_pdf(out_dir);
      direct = (direct * s1.atten) * (1.0f / pdf);
    }
    radiance = radiance + direct;
    if (!discrete) {
      radiance = radiance - direct;
      const V3 to_light = light_sample(S, sf.position, rng);
      const V3 chosen = rng.coin(0.5f) ? world_in : to_light;
      const Ray r6 = make_ray(sf.position, chosen, kEps, FLT_MAX);
      // the ray-log coin is always flipped (student/pathtracer.cpp:148); when it fires the ray goes to the GUI's log
      if (rng.coin(0.0005f) && S.ray_log) log_ray_event(S.ray_log, S.ray_log_cap, r6.o.x, r6.o.y, r6.o.z, r6.d.x, r6.d.y, r6.d.z, rng.inc, (uint32_t)level);
      Spec d6 = emitted_along<COUNT>(S, r6, cnt);
      const float pdf_area = light_pdf<COUNT>(S, sf.position, to_light, cnt);
      const float pdf4 = lambert_pdf(out_dir);
      pdf = (pdf4 + pdf_area) / 2.0f;
      const Spec att6 = lambert_evaluate(m, out_dir);
      d6 = (d6 * att6) * (1.0f / pdf);
      radiance = radiance + d6;
    }

    // ---- sample_indirect_lighting: scatter again, recurse with depth - 1 ----
    const Scatter s2 = scatter(m, out_dir, rng);
    const V3 world_in2 = frame_to_world(fr, s2.dir);
    Bounce& b = rec[level];
    b.direct = radiance;
    b.atten = s2.atten;
    b.discrete = discrete ? 1u : 0u;
    b.inv_pdf = discrete ? 0.0f : (1.0f / lambert_pdf(out_dir));
    level++;
    ray = make_ray(sf.position, world_in2, kEps, FLT_MAX);
    depth = depth - 1;
  }
  // Unwind: every terminal trace() has .second == 0.
  Spec L = spec(0, 0, 0);
  for (int k = level - 1; k >= 0; k--) {
    const Bounce& b = rec[k];
    Spec ind = b.discrete ? (L * b.atten) : ((L * b.atten) * b.inv_pdf);
    ind = spec(0, 0, 0) + ind;   // `radiance += indirect_light` on a zero Spectrum
    L = b.direct + ind;
  }
  return emissive_cam + L;
}

}  // namespace srt

#endif
