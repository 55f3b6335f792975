// MI355X (gfx950) path tracer: Pathtracer::trace_pixel and everything below it as HIP kernels, plus the
// C ABI of include/srt_pt.h.  Reference paths are relative to /root/reference/Assignments/Scotty3D/src/.
//
//   trace_pixel               student/pathtracer.cpp:14-40      -> path_sample()
//   trace                     student/pathtracer.cpp:174-218    -> the bounce loop of path_sample()
//   sample_direct_lighting    student/pathtracer.cpp:78-172     -> direct block (BSDF ray + MIS ray)
//   sample_indirect_lighting  student/pathtracer.cpp:42-76      -> per-bounce record folded bottom-up
//   BVH<>::hit                student/bvh.inl:166-276           -> traverse<>() (explicit-stack form of the recursion)
//   Object::hit               rays/object.h:57-65               -> object_hit()
//   sample_area_lights / area_lights_pdf  rays/pathtracer.cpp:301-325 -> light_sample() / light_pdf()
//
// The reference recursion `L = direct + (L_next * atten) * (1/pdf)` is evaluated leaf-first; to round
// identically the kernel records (direct, atten, 1/pdf) per bounce and folds the records from the last
// bounce back to the first instead of carrying a running throughput.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "pt_device.h"
#include "pt_scene.h"
#include "srt_common.h"
#include "srt_pt.h"

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
  uint32_t nobjects, nlights, tlas_nodes, use_bvh, light_tri_first;
  Camera cam;
  uint32_t w, h, max_depth;
};

enum { C_RAYS = 0, C_BOX, C_OBJ, C_TRI, C_SPH, C_TLAS, C_BLAS, C_LTRI, C_COUNT };
struct Counters { uint32_t v[C_COUNT]; };

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

// Closest triangle of one mesh in OBJECT space: Tri_Mesh::hit -> BVH<Triangle>::hit / List<Triangle>::hit.
// Returns ids in Hit (tri = global triangle index) and the (u, v, t) of the winner through uvt.
template <bool COUNT>
SRT_DEV Hit mesh_hit(const DScene& S, const Object& o, const Ray& oray, Counters& cnt) {
  Hit best; best.hit = false; best.dist = 0.0f; best.obj = 0; best.tri = 0;
  if (o.use_bvh) {
    if (o.nnodes == 0) return best;
    const float dn = norm(oray.d);
    const float tx = oray.b0 / dn, ty = oray.b1 / dn;  // Vec2 time_initial = dist_bounds / dir.norm()
    auto leaf = [&](uint32_t slot, Hit& acc) {
      if (COUNT) cnt.v[C_TRI]++;
      const TriHit th = tri_hit(S.tris[o.tri_base + slot], oray);
      fold(acc, th.hit, th.dist, 0, o.tri_base + slot);
    };
    return traverse<kMaxBlasDepth, COUNT>(S.nodes + o.node_base, oray, tx, ty, cnt, C_BLAS, leaf);
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
SRT_DEV Scatter scatter(const Material& m, V3 out, Rng& rng) {
  Scatter r;
  if (m.type == 0) {                                   // BSDF_Lambertian::scatter, bsdf.cpp:69-87
    const float phi = rng.unit() * 2.0f * kPi;         // Hemisphere::Cosine::sample, samplers.cpp:166-177
    const float cos_t = sqrtf(rng.unit());
    const float sin_t = sqrtf(1 - cos_t * cos_t);
    const float x = srt_cosf(phi) * sin_t;
    const float z = srt_sinf(phi) * sin_t;
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

// Pathtracer::sample_area_lights (no environment light on this path): List<Object>::sample ->
// Object::sample -> List<Triangle>::sample -> Samplers::Triangle::sample.
SRT_DEV V3 light_sample(const DScene& S, V3 from, Rng& rng) {
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
      for (uint32_t t = 0; t < L.ntri; t++) {
        if (COUNT) cnt.v[C_LTRI]++;
        Ray tray = wray;
        ray_transform(tray, L.pdfiT);           // applied even when iT is the identity
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
  if (n) pdf /= n;
  return pdf;
}

// Camera::generate_ray (student/camera.cpp:7-34); screen_h/screen_w come from the host (tanf).
SRT_DEV Ray camera_ray(const DScene& S, float sx, float sy) {
  const float sh = S.cam.screen_h, sw = S.cam.screen_w;
  Ray r;
  r.o = v3(0, 0, 0);
  r.d = v3(sx * sw - 0.5f * sw, sy * sh - 0.5f * sh, -1.0f);
  r.b0 = 0.0f;
  r.b1 = __uint_as_float(0x7f800000u);
  ray_transform(r, S.cam.iview);
  return r;
}

// `trace(ray).first` of a depth-0 ray: emitted radiance of whatever it hits, else zero.
template <bool COUNT>
SRT_DEV Spec emitted_along(const DScene& S, const Ray& ray, Counters& cnt) {
  const Hit h = scene_hit<COUNT>(S, ray, cnt);
  if (!h.hit) return spec(0, 0, 0);
  const Spec e = emissive_of(S.materials[S.objects[h.obj].material]);
  return (luma(e) > 0.0f) ? e : spec(0, 0, 0);
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
    if (!h.hit) break;
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
    Spec radiance = spec(0, 0, 0);  // point_lighting(): no delta lights on this path
    const Scatter s1 = scatter(m, out_dir, rng);
    const V3 world_in = frame_to_world(fr, s1.dir);
    const Ray r1 = make_ray(sf.position, world_in, kEps, FLT_MAX);
    Spec direct = emitted_along<COUNT>(S, r1, cnt);
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
      (void)rng.coin(0.0005f);  // the ray-log coin is always flipped (student/pathtracer.cpp:148)
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

// ---------------------------------------------------------------------------------------------------
// Kernels
// ---------------------------------------------------------------------------------------------------
struct TileMap { uint32_t tile_w, tile_h, tiles_x, tiles_y, rank, world, local_tiles; };

// One lane per pixel of this rank's tiles; the lane walks the epoch's samples in order so the per-pixel
// sum is accumulated exactly like do_trace (rays/pathtracer.cpp:216-226).
__global__ __launch_bounds__(64) void pt_epoch_kernel(DScene S, TileMap T, uint64_t seed, uint32_t sample_base,
                                                      uint32_t samples, float* __restrict__ tiles_out,
                                                      unsigned long long* __restrict__ ray_counter) {
  const uint32_t px_per_tile = T.tile_w * T.tile_h;
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t local_tile = gid / px_per_tile;
  const uint32_t in_tile = gid % px_per_tile;
  if (local_tile >= T.local_tiles) return;
  // 8x8 pixel blocks per wavefront inside the tile: neighbouring pixels take similar paths
  const uint32_t blocks_x = T.tile_w / 8;
  const uint32_t blk = in_tile / 64, lane = in_tile % 64;
  const uint32_t lx = (blk % blocks_x) * 8 + (lane % 8), ly = (blk / blocks_x) * 8 + (lane / 8);
  const uint32_t tile = T.rank + local_tile * T.world;
  const uint32_t x = (tile % T.tiles_x) * T.tile_w + lx, y = (tile / T.tiles_x) * T.tile_h + ly;
  float* out = tiles_out + ((size_t)local_tile * px_per_tile + (size_t)ly * T.tile_w + lx) * 3;
  Counters cnt;
  cnt.v[C_RAYS] = 0;
  const bool inside = x < S.w && y < S.h;
  if (!inside) samples = 0;  // padding lanes of edge tiles: no samples, zero output, still join the wave reduction
  Rng rng;
  Spec acc = spec(0, 0, 0);
  uint32_t sampled = 0;
  for (uint32_t s = 0; s < samples; s++) {
    rng.key(seed, y * S.w + x, sample_base + s);
    const Spec p = path_sample<false>(S, x, y, rng, cnt);
    if (valid(p)) { acc = acc + p; sampled++; }
  }
  if (sampled > 0) acc = acc * (1.0f / sampled);
  out[0] = acc.r; out[1] = acc.g; out[2] = acc.b;
  if (ray_counter) {  // one atomic per wavefront (every lane of the wave reaches this point)
    unsigned long long r = cnt.v[C_RAYS];
    for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(ray_counter, r);
  }
}

// Explicit (x, y, sample) triples; instrumented when COUNT.
template <bool COUNT>
__global__ __launch_bounds__(64) void pt_samples_kernel(DScene S, uint64_t seed, const uint32_t* __restrict__ xs,
                                                        const uint32_t* __restrict__ ys, const uint32_t* __restrict__ ss,
                                                        uint32_t n, float* __restrict__ rgb, uint32_t* __restrict__ draws,
                                                        uint32_t* __restrict__ rays, unsigned long long* __restrict__ totals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Counters cnt;
  for (int k = 0; k < C_COUNT; k++) cnt.v[k] = 0;
  Rng rng;
  rng.key(seed, ys[i] * S.w + xs[i], ss[i]);
  const Spec p = path_sample<COUNT>(S, xs[i], ys[i], rng, cnt);
  rgb[3 * i] = p.r; rgb[3 * i + 1] = p.g; rgb[3 * i + 2] = p.b;
  if (draws) draws[i] = rng.draws;
  if (COUNT) {
    if (rays) rays[i] = cnt.v[C_RAYS];
    for (int k = 0; k < C_COUNT; k++) atomicAdd(&totals[k], (unsigned long long)cnt.v[k]);
  }
}

__global__ void pt_hit_kernel(DScene S, const float* __restrict__ org, const float* __restrict__ dir,
                              const float* __restrict__ bounds, uint32_t n, float* __restrict__ out9) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ray r;
  r.o = v3(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
  r.d = v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
  r.b0 = bounds[2 * i]; r.b1 = bounds[2 * i + 1];
  Counters cnt;
  const Hit h = scene_hit<false>(S, r, cnt);
  float* o = out9 + 9 * i;
  for (int k = 0; k < 9; k++) o[k] = 0.0f;
  if (h.hit) {
    const Surface sf = surface_of(S, h, r);
    o[0] = 1.0f; o[1] = h.dist;
    o[2] = sf.position.x; o[3] = sf.position.y; o[4] = sf.position.z;
    o[5] = sf.normal.x; o[6] = sf.normal.y; o[7] = sf.normal.z;
    o[8] = (float)S.objects[h.obj].material;
  }
}

__global__ void pt_untile_kernel(TileMap T, uint32_t w, uint32_t h, uint32_t tiles_per_rank,
                                 const float* __restrict__ gathered, float* __restrict__ image) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  const uint32_t x = i % w, y = i / w;
  const uint32_t tile = (y / T.tile_h) * T.tiles_x + (x / T.tile_w);
  const uint32_t rank = tile % T.world, local = tile / T.world;
  const size_t src = (((size_t)rank * tiles_per_rank + local) * (T.tile_w * T.tile_h) + (size_t)(y % T.tile_h) * T.tile_w +
                      (x % T.tile_w)) * 3;
  image[3 * (size_t)i] = gathered[src];
  image[3 * (size_t)i + 1] = gathered[src + 1];
  image[3 * (size_t)i + 2] = gathered[src + 2];
}

// Pathtracer::accumulate: s += (n - s) * (1.0f / accumulator_samples)
__global__ void pt_accumulate_kernel(float* __restrict__ acc, const float* __restrict__ epoch, size_t n, float inv) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) acc[i] += (epoch[i] - acc[i]) * inv;
}

__global__ void pt_math_kernel(const float* __restrict__ x, size_t n, float* __restrict__ c, float* __restrict__ s) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { c[i] = srt_cosf(x[i]); s[i] = srt_sinf(x[i]); }
}

}  // namespace srt

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
using namespace srt;

struct srt_pt {
  int device = -1;            // -1: host-only context (scene assembly / BVH inspection, no rendering)
  hipStream_t stream = nullptr;
  std::vector<ObjectInput> inputs;
  std::vector<Material> materials;
  BuiltScene built;
  bool committed = false;
  Camera cam{};
  bool have_cam = false;
  uint32_t w = 0, h = 0, max_depth = 8;
  TileMap tiles{32, 32, 0, 0, 0, 1, 0};
  uint32_t tiles_per_rank = 0;
  // device copies
  Node* d_nodes = nullptr; Tri* d_tris = nullptr; TriNrm* d_nrm = nullptr; Object* d_objects = nullptr;
  Light* d_lights = nullptr; LightTri* d_ltris = nullptr; Material* d_mats = nullptr;
  float* d_tile_buf = nullptr; size_t tile_buf_floats = 0;
  float* d_image = nullptr; size_t image_floats = 0;
  unsigned long long* d_totals = nullptr;   // C_COUNT instrumented totals + 1 slot: rays of the epoch kernels
  unsigned long long last_counters[C_COUNT] = {0};
  uint64_t camera_samples = 0;
};

namespace {

template <typename T>
int upload(T** dst, const std::vector<T>& src) {
  if (*dst) { SRT_HIP(hipFree(*dst)); *dst = nullptr; }
  const size_t n = src.empty() ? 1 : src.size();
  SRT_HIP(hipMalloc(dst, n * sizeof(T)));
  if (!src.empty()) SRT_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return SRT_OK;
}

int need_device(srt_pt* pt, const char* what) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "%s: NULL context", what);
  if (pt->device < 0) return srt::fail(SRT_ERR_NO_DEVICE, "%s needs a HIP device; this context is host-only and there is no CPU fallback", what);
  SRT_HIP(hipSetDevice(pt->device));
  return SRT_OK;
}

int need_ready(srt_pt* pt, const char* what) {
  int st = need_device(pt, what);
  if (st != SRT_OK) return st;
  if (!pt->committed) return srt::fail(SRT_ERR_STATE, "%s before srt_pt_scene_commit", what);
  if (!pt->have_cam) return srt::fail(SRT_ERR_STATE, "%s before srt_pt_set_camera", what);
  if (!pt->w || !pt->h) return srt::fail(SRT_ERR_STATE, "%s before srt_pt_set_params", what);
  return SRT_OK;
}

DScene device_scene(const srt_pt* pt) {
  const FlatScene& F = pt->built.flat;
  DScene S;
  S.nodes = pt->d_nodes; S.tris = pt->d_tris; S.tri_nrm = pt->d_nrm; S.objects = pt->d_objects;
  S.lights = pt->d_lights; S.light_tris = pt->d_ltris; S.materials = pt->d_mats;
  S.nobjects = (uint32_t)F.objects.size(); S.nlights = (uint32_t)F.lights.size();
  S.tlas_nodes = F.tlas_nodes; S.use_bvh = F.use_bvh ? 1u : 0u; S.light_tri_first = F.light_tri_first;
  S.cam = pt->cam; S.w = pt->w; S.h = pt->h; S.max_depth = pt->max_depth;
  return S;
}

void update_tiling(srt_pt* pt) {
  TileMap& T = pt->tiles;
  if (!pt->w || !pt->h) { T.tiles_x = T.tiles_y = T.local_tiles = 0; pt->tiles_per_rank = 0; return; }
  T.tiles_x = (pt->w + T.tile_w - 1) / T.tile_w;
  T.tiles_y = (pt->h + T.tile_h - 1) / T.tile_h;
  const uint32_t ntiles = T.tiles_x * T.tiles_y;
  pt->tiles_per_rank = (ntiles + T.world - 1) / T.world;
  T.local_tiles = (ntiles > T.rank) ? (ntiles - T.rank + T.world - 1) / T.world : 0;
}

}  // namespace

extern "C" {

int srt_pt_create(int device, srt_pt** out) {
  if (!out) return srt::fail(SRT_ERR_INVALID, "srt_pt_create: out is NULL");
  *out = nullptr;
  srt_pt* pt = new (std::nothrow) srt_pt();
  if (!pt) return srt::fail(SRT_ERR_INVALID, "out of host memory");
  if (device >= 0) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
      delete pt;
      return srt::fail(SRT_ERR_NO_DEVICE, "no HIP device available (%s); this path has no CPU fallback",
                       e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    if (device >= count) { delete pt; return srt::fail(SRT_ERR_INVALID, "device %d out of range [0,%d)", device, count); }
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&pt->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&pt->d_totals, (C_COUNT + 1) * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(pt->d_totals, 0, (C_COUNT + 1) * sizeof(unsigned long long)) != hipSuccess) {
      delete pt;
      return srt::fail(SRT_ERR_HIP, "HIP context setup failed on device %d", device);
    }
    pt->device = device;
  }
  *out = pt;
  return SRT_OK;
}

int srt_pt_destroy(srt_pt* pt) {
  if (!pt) return SRT_OK;
  if (pt->device >= 0) {
    (void)hipSetDevice(pt->device);
    (void)hipStreamSynchronize(pt->stream);
    (void)hipFree(pt->d_nodes); (void)hipFree(pt->d_tris); (void)hipFree(pt->d_nrm); (void)hipFree(pt->d_objects);
    (void)hipFree(pt->d_lights); (void)hipFree(pt->d_ltris); (void)hipFree(pt->d_mats);
    (void)hipFree(pt->d_tile_buf); (void)hipFree(pt->d_image); (void)hipFree(pt->d_totals);
    (void)hipStreamDestroy(pt->stream);
  }
  delete pt;
  return SRT_OK;
}

int srt_pt_scene_begin(srt_pt* pt) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_scene_begin: NULL context");
  pt->inputs.clear();
  pt->materials.clear();
  pt->committed = false;
  return SRT_OK;
}

int srt_pt_add_material(srt_pt* pt, const srt_pt_material* m, uint32_t* index_out) {
  if (!pt || !m) return srt::fail(SRT_ERR_INVALID, "srt_pt_add_material: NULL argument");
  if (m->type > SRT_MAT_REFRACT) return srt::fail(SRT_ERR_INVALID, "unknown material type %u", m->type);
  if (pt->committed) return srt::fail(SRT_ERR_STATE, "scene already committed; call srt_pt_scene_begin first");
  Material mm;
  mm.type = m->type;
  for (int i = 0; i < 3; i++) { mm.a[i] = m->a[i]; mm.b[i] = m->b[i]; }
  mm.ior = m->ior;
  pt->materials.push_back(mm);
  if (index_out) *index_out = (uint32_t)pt->materials.size() - 1;
  return SRT_OK;
}

int srt_pt_add_mesh(srt_pt* pt, const float* positions, const float* normals, uint32_t nverts, const uint32_t* indices,
                    uint32_t nindices, const float trans[16], uint32_t material, int is_area_light) {
  if (!pt || !positions || !normals || !indices || !trans) return srt::fail(SRT_ERR_INVALID, "srt_pt_add_mesh: NULL argument");
  if (pt->committed) return srt::fail(SRT_ERR_STATE, "scene already committed; call srt_pt_scene_begin first");
  if (nverts == 0 || nindices == 0 || nindices % 3) return srt::fail(SRT_ERR_INVALID, "mesh needs >= 1 triangle (nindices %u)", nindices);
  if (material >= pt->materials.size()) return srt::fail(SRT_ERR_INVALID, "material %u not defined", material);
  for (uint32_t i = 0; i < nindices; i++)
    if (indices[i] >= nverts) return srt::fail(SRT_ERR_INVALID, "index %u (= %u) out of range (nverts %u)", i, indices[i], nverts);
  ObjectInput o;
  o.kind = OBJ_MESH;
  std::memcpy(&o.trans, trans, sizeof(Mat4));
  o.material = material;
  o.is_light = is_area_light != 0;
  o.mesh.pos.assign(positions, positions + 3 * (size_t)nverts);
  o.mesh.nrm.assign(normals, normals + 3 * (size_t)nverts);
  o.mesh.idx.assign(indices, indices + nindices);
  pt->inputs.push_back(std::move(o));
  return SRT_OK;
}

int srt_pt_add_sphere(srt_pt* pt, float radius, const float trans[16], uint32_t material) {
  if (!pt || !trans) return srt::fail(SRT_ERR_INVALID, "srt_pt_add_sphere: NULL argument");
  if (pt->committed) return srt::fail(SRT_ERR_STATE, "scene already committed; call srt_pt_scene_begin first");
  if (material >= pt->materials.size()) return srt::fail(SRT_ERR_INVALID, "material %u not defined", material);
  ObjectInput o;
  o.kind = OBJ_SPHERE;
  std::memcpy(&o.trans, trans, sizeof(Mat4));
  o.material = material;
  o.radius = radius;
  pt->inputs.push_back(std::move(o));
  return SRT_OK;
}

int srt_pt_scene_commit(srt_pt* pt, int use_bvh) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_scene_commit: NULL context");
  const std::string err = build_scene(pt->inputs, pt->materials, use_bvh != 0, &pt->built);
  if (!err.empty()) return srt::fail(SRT_ERR_UNSUPPORTED, "%s", err.c_str());
  const FlatScene& F = pt->built.flat;
  if ((int)F.max_tlas_depth > kMaxTlasDepth || (int)F.max_blas_depth > kMaxBlasDepth)
    return srt::fail(SRT_ERR_UNSUPPORTED, "BVH too deep for the traversal stacks (TLAS %u > %d or BLAS %u > %d)",
                     F.max_tlas_depth, kMaxTlasDepth, F.max_blas_depth, kMaxBlasDepth);
  if (pt->device >= 0) {
    SRT_HIP(hipSetDevice(pt->device));
    SRT_HIP(hipStreamSynchronize(pt->stream));
    int st;
    if ((st = upload(&pt->d_nodes, F.nodes)) || (st = upload(&pt->d_tris, F.tris)) || (st = upload(&pt->d_nrm, F.tri_nrm)) ||
        (st = upload(&pt->d_objects, F.objects)) || (st = upload(&pt->d_lights, F.lights)) ||
        (st = upload(&pt->d_ltris, F.light_tris)) || (st = upload(&pt->d_mats, F.materials)))
      return st;
  }
  pt->committed = true;
  return SRT_OK;
}

int srt_pt_set_camera(srt_pt* pt, const float iview[16], float vert_fov_deg, float aspect_ratio) {
  if (!pt || !iview) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_camera: NULL argument");
  pt->cam = make_camera(iview, vert_fov_deg, aspect_ratio);
  pt->have_cam = true;
  return SRT_OK;
}

int srt_pt_set_params(srt_pt* pt, uint32_t width, uint32_t height, uint32_t max_depth) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_params: NULL context");
  if (!width || !height) return srt::fail(SRT_ERR_INVALID, "image must be at least 1x1 (got %ux%u)", width, height);
  if ((uint64_t)width * height > 0x7fffffffull) return srt::fail(SRT_ERR_UNSUPPORTED, "image larger than 2^31 pixels");
  if (max_depth > (uint32_t)kMaxPathDepth) return srt::fail(SRT_ERR_UNSUPPORTED, "max_depth %u > %d is not supported", max_depth, kMaxPathDepth);
  pt->w = width; pt->h = height; pt->max_depth = max_depth;
  update_tiling(pt);
  return SRT_OK;
}

int srt_pt_set_tiling(srt_pt* pt, uint32_t tile_w, uint32_t tile_h, uint32_t rank, uint32_t world) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_tiling: NULL context");
  if (!tile_w || !tile_h || tile_w % 8 || tile_h % 8) return srt::fail(SRT_ERR_INVALID, "tile size must be a positive multiple of 8 (got %ux%u)", tile_w, tile_h);
  if (!world || rank >= world) return srt::fail(SRT_ERR_INVALID, "rank %u / world %u is not a valid shard", rank, world);
  pt->tiles.tile_w = tile_w; pt->tiles.tile_h = tile_h; pt->tiles.rank = rank; pt->tiles.world = world;
  update_tiling(pt);
  return SRT_OK;
}

int srt_pt_tile_info(srt_pt* pt, uint32_t* local_tiles, uint32_t* tiles_per_rank, uint32_t* floats_per_tile) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_tile_info: NULL context");
  if (!pt->w) return srt::fail(SRT_ERR_STATE, "srt_pt_tile_info before srt_pt_set_params");
  if (local_tiles) *local_tiles = pt->tiles.local_tiles;
  if (tiles_per_rank) *tiles_per_rank = pt->tiles_per_rank;
  if (floats_per_tile) *floats_per_tile = pt->tiles.tile_w * pt->tiles.tile_h * 3;
  return SRT_OK;
}

int srt_pt_render_epoch_device(srt_pt* pt, void* stream, uint64_t seed, uint32_t sample_base, uint32_t samples,
                               float* d_tiles_out) {
  int st = need_ready(pt, "srt_pt_render_epoch_device");
  if (st != SRT_OK) return st;
  if (!d_tiles_out) return srt::fail(SRT_ERR_INVALID, "srt_pt_render_epoch_device: output is NULL");
  hipStream_t s = (hipStream_t)stream;  // exactly the caller's stream; NULL is the HIP default stream
  const TileMap& T = pt->tiles;
  const uint64_t lanes = (uint64_t)T.local_tiles * T.tile_w * T.tile_h;
  if (lanes) {
    const uint32_t blocks = (uint32_t)((lanes + 63) / 64);
    pt_epoch_kernel<<<dim3(blocks), dim3(64), 0, s>>>(device_scene(pt), T, seed, sample_base, samples, d_tiles_out,
                                                      pt->d_totals + C_COUNT);
    SRT_HIP(hipGetLastError());
    // camera samples of this epoch: pixels of this rank's tiles that lie inside the image
    uint64_t px = 0;
    for (uint32_t k = 0; k < T.local_tiles; k++) {
      const uint32_t tile = T.rank + k * T.world;
      const uint32_t x0 = (tile % T.tiles_x) * T.tile_w, y0 = (tile / T.tiles_x) * T.tile_h;
      px += (uint64_t)std::min(T.tile_w, pt->w - x0) * std::min(T.tile_h, pt->h - y0);
    }
    pt->camera_samples += px * samples;
  }
  return SRT_OK;
}

int srt_pt_untile_device(srt_pt* pt, void* stream, const float* d_gathered, float* d_image) {
  int st = need_ready(pt, "srt_pt_untile_device");
  if (st != SRT_OK) return st;
  if (!d_gathered || !d_image) return srt::fail(SRT_ERR_INVALID, "srt_pt_untile_device: NULL buffer");
  hipStream_t s = (hipStream_t)stream;  // exactly the caller's stream; NULL is the HIP default stream
  const uint32_t n = pt->w * pt->h;
  pt_untile_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(pt->tiles, pt->w, pt->h, pt->tiles_per_rank, d_gathered, d_image);
  SRT_HIP(hipGetLastError());
  return SRT_OK;
}

int srt_pt_accumulate_device(srt_pt* pt, void* stream, float* d_accumulator, const float* d_epoch, size_t nfloats,
                             uint32_t accumulator_samples) {
  int st = need_device(pt, "srt_pt_accumulate_device");
  if (st != SRT_OK) return st;
  if (!d_accumulator || !d_epoch || !accumulator_samples) return srt::fail(SRT_ERR_INVALID, "srt_pt_accumulate_device: bad argument");
  hipStream_t s = (hipStream_t)stream;  // exactly the caller's stream; NULL is the HIP default stream
  pt_accumulate_kernel<<<dim3((unsigned)((nfloats + 255) / 256)), dim3(256), 0, s>>>(d_accumulator, d_epoch, nfloats,
                                                                                 1.0f / accumulator_samples);
  SRT_HIP(hipGetLastError());
  return SRT_OK;
}

int srt_pt_render_epoch(srt_pt* pt, uint64_t seed, uint32_t sample_base, uint32_t samples, float* rgb_out) {
  int st = need_ready(pt, "srt_pt_render_epoch");
  if (st != SRT_OK) return st;
  if (!rgb_out) return srt::fail(SRT_ERR_INVALID, "srt_pt_render_epoch: output is NULL");
  const TileMap& T = pt->tiles;
  const size_t per_tile = (size_t)T.tile_w * T.tile_h * 3;
  const size_t need = per_tile * (T.local_tiles ? T.local_tiles : 1);
  if (pt->tile_buf_floats < need) {
    if (pt->d_tile_buf) SRT_HIP(hipFree(pt->d_tile_buf));
    pt->d_tile_buf = nullptr;
    SRT_HIP(hipMalloc(&pt->d_tile_buf, need * sizeof(float)));
    pt->tile_buf_floats = need;
  }
  st = srt_pt_render_epoch_device(pt, (void*)pt->stream, seed, sample_base, samples, pt->d_tile_buf);
  if (st != SRT_OK) return st;
  std::vector<float> host(per_tile * T.local_tiles);
  if (!host.empty()) SRT_HIP(hipMemcpyAsync(host.data(), pt->d_tile_buf, host.size() * sizeof(float), hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipStreamSynchronize(pt->stream));
  for (uint32_t k = 0; k < T.local_tiles; k++) {
    const uint32_t tile = T.rank + k * T.world;
    const uint32_t x0 = (tile % T.tiles_x) * T.tile_w, y0 = (tile / T.tiles_x) * T.tile_h;
    for (uint32_t ly = 0; ly < T.tile_h && y0 + ly < pt->h; ly++)
      for (uint32_t lx = 0; lx < T.tile_w && x0 + lx < pt->w; lx++) {
        const float* src = &host[k * per_tile + ((size_t)ly * T.tile_w + lx) * 3];
        float* dst = rgb_out + ((size_t)(y0 + ly) * pt->w + (x0 + lx)) * 3;
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
      }
  }
  return SRT_OK;
}

int srt_pt_ray_count(srt_pt* pt, uint64_t* rays, uint64_t* camera_samples, int reset) {
  int st = need_device(pt, "srt_pt_ray_count");
  if (st != SRT_OK) return st;
  SRT_HIP(hipDeviceSynchronize());  // epochs may be in flight on any stream
  unsigned long long r = 0;
  SRT_HIP(hipMemcpy(&r, pt->d_totals + C_COUNT, sizeof r, hipMemcpyDeviceToHost));
  if (rays) *rays = r;
  if (camera_samples) *camera_samples = pt->camera_samples;
  if (reset) {
    SRT_HIP(hipMemset(pt->d_totals + C_COUNT, 0, sizeof r));
    pt->camera_samples = 0;
  }
  return SRT_OK;
}

int srt_pt_trace_samples(srt_pt* pt, uint64_t seed, const uint32_t* xs, const uint32_t* ys, const uint32_t* ss, size_t n,
                         float* rgb_out, uint32_t* draws_out, uint32_t* rays_out) {
  int st = need_ready(pt, "srt_pt_trace_samples");
  if (st != SRT_OK) return st;
  if (n == 0) return SRT_OK;
  if (!xs || !ys || !ss || !rgb_out) return srt::fail(SRT_ERR_INVALID, "srt_pt_trace_samples: NULL argument");
  if (n > 0x7fffffffull) return srt::fail(SRT_ERR_UNSUPPORTED, "too many samples in one call");
  for (size_t i = 0; i < n; i++)
    if (xs[i] >= pt->w || ys[i] >= pt->h) return srt::fail(SRT_ERR_INVALID, "sample %zu: pixel (%u,%u) outside %ux%u", i, xs[i], ys[i], pt->w, pt->h);
  uint32_t *dx = nullptr, *dy = nullptr, *ds = nullptr, *dd = nullptr, *dr = nullptr;
  float* drgb = nullptr;
  SRT_HIP(hipMalloc(&dx, n * 4)); SRT_HIP(hipMalloc(&dy, n * 4)); SRT_HIP(hipMalloc(&ds, n * 4));
  SRT_HIP(hipMalloc(&dd, n * 4)); SRT_HIP(hipMalloc(&dr, n * 4)); SRT_HIP(hipMalloc(&drgb, n * 12));
  SRT_HIP(hipMemcpyAsync(dx, xs, n * 4, hipMemcpyHostToDevice, pt->stream));
  SRT_HIP(hipMemcpyAsync(dy, ys, n * 4, hipMemcpyHostToDevice, pt->stream));
  SRT_HIP(hipMemcpyAsync(ds, ss, n * 4, hipMemcpyHostToDevice, pt->stream));
  SRT_HIP(hipMemsetAsync(pt->d_totals, 0, C_COUNT * sizeof(unsigned long long), pt->stream));
  pt_samples_kernel<true><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, pt->stream>>>(device_scene(pt), seed, dx, dy, ds, (uint32_t)n,
                                                                                   drgb, dd, dr, pt->d_totals);
  SRT_HIP(hipGetLastError());
  SRT_HIP(hipMemcpyAsync(rgb_out, drgb, n * 12, hipMemcpyDeviceToHost, pt->stream));
  if (draws_out) SRT_HIP(hipMemcpyAsync(draws_out, dd, n * 4, hipMemcpyDeviceToHost, pt->stream));
  if (rays_out) SRT_HIP(hipMemcpyAsync(rays_out, dr, n * 4, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipMemcpyAsync(pt->last_counters, pt->d_totals, sizeof pt->last_counters, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipStreamSynchronize(pt->stream));
  (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(ds); (void)hipFree(dd); (void)hipFree(dr); (void)hipFree(drgb);
  return SRT_OK;
}

int srt_pt_hit(srt_pt* pt, const float* origins, const float* dirs, const float* bounds, size_t n, float* out9) {
  int st = need_device(pt, "srt_pt_hit");
  if (st != SRT_OK) return st;
  if (!pt->committed) return srt::fail(SRT_ERR_STATE, "srt_pt_hit before srt_pt_scene_commit");
  if (n == 0) return SRT_OK;
  if (!origins || !dirs || !bounds || !out9) return srt::fail(SRT_ERR_INVALID, "srt_pt_hit: NULL argument");
  float *dorg = nullptr, *ddir = nullptr, *db = nullptr, *dout = nullptr;
  SRT_HIP(hipMalloc(&dorg, n * 12)); SRT_HIP(hipMalloc(&ddir, n * 12)); SRT_HIP(hipMalloc(&db, n * 8)); SRT_HIP(hipMalloc(&dout, n * 36));
  SRT_HIP(hipMemcpyAsync(dorg, origins, n * 12, hipMemcpyHostToDevice, pt->stream));
  SRT_HIP(hipMemcpyAsync(ddir, dirs, n * 12, hipMemcpyHostToDevice, pt->stream));
  SRT_HIP(hipMemcpyAsync(db, bounds, n * 8, hipMemcpyHostToDevice, pt->stream));
  DScene S = device_scene(pt);
  pt_hit_kernel<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, pt->stream>>>(S, dorg, ddir, db, (uint32_t)n, dout);
  SRT_HIP(hipGetLastError());
  SRT_HIP(hipMemcpyAsync(out9, dout, n * 36, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipStreamSynchronize(pt->stream));
  (void)hipFree(dorg); (void)hipFree(ddir); (void)hipFree(db); (void)hipFree(dout);
  return SRT_OK;
}

long srt_pt_dump_bvh(srt_pt* pt, int which, float* boxes, uint32_t* links, size_t cap, uint32_t* order) {
  if (!pt || !boxes || !links) return srt::fail(SRT_ERR_INVALID, "srt_pt_dump_bvh: NULL argument");
  if (!pt->committed) return srt::fail(SRT_ERR_STATE, "srt_pt_dump_bvh before srt_pt_scene_commit");
  if (!pt->built.flat.use_bvh) return srt::fail(SRT_ERR_STATE, "scene was committed without BVHs");
  const HostBVH* b = &pt->built.tlas;
  const ObjectInput* in = nullptr;
  if (which >= 0) {
    if ((size_t)which >= pt->built.tlas.prim.size()) return srt::fail(SRT_ERR_INVALID, "object slot %d out of range", which);
    const uint32_t obj = pt->built.tlas.prim[which];
    in = &pt->built.inputs[obj];
    if (in->kind != OBJ_MESH) return srt::fail(SRT_ERR_INVALID, "object slot %d is not a mesh", which);
    b = &pt->built.blas[obj];
  }
  for (size_t i = 0; i < b->nodes.size() && i < cap; i++) {
    const HostNode& nd = b->nodes[i];
    for (int a = 0; a < 3; a++) { boxes[6 * i + a] = nd.mn[a]; boxes[6 * i + 3 + a] = nd.mx[a]; }
    links[4 * i] = nd.start; links[4 * i + 1] = nd.size; links[4 * i + 2] = nd.l; links[4 * i + 3] = nd.r;
  }
  if (order)
    for (size_t i = 0; i < b->prim.size(); i++) order[i] = in ? in->mesh.idx[3 * b->prim[i]] : b->prim[i] + 1;
  return (long)b->nodes.size();
}

int srt_pt_counters(srt_pt* pt, uint64_t out[8]) {
  if (!pt || !out) return srt::fail(SRT_ERR_INVALID, "srt_pt_counters: NULL argument");
  for (int k = 0; k < C_COUNT; k++) out[k] = pt->last_counters[k];
  return SRT_OK;
}

int srt_pt_math_cos_sin(srt_pt* pt, const float* x, size_t n, float* cos_out, float* sin_out) {
  int st = need_device(pt, "srt_pt_math_cos_sin");
  if (st != SRT_OK) return st;
  if (n == 0) return SRT_OK;
  float *dx = nullptr, *dc = nullptr, *dsn = nullptr;
  SRT_HIP(hipMalloc(&dx, n * 4)); SRT_HIP(hipMalloc(&dc, n * 4)); SRT_HIP(hipMalloc(&dsn, n * 4));
  SRT_HIP(hipMemcpyAsync(dx, x, n * 4, hipMemcpyHostToDevice, pt->stream));
  pt_math_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pt->stream>>>(dx, n, dc, dsn);
  SRT_HIP(hipGetLastError());
  SRT_HIP(hipMemcpyAsync(cos_out, dc, n * 4, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipMemcpyAsync(sin_out, dsn, n * 4, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipStreamSynchronize(pt->stream));
  (void)hipFree(dx); (void)hipFree(dc); (void)hipFree(dsn);
  return SRT_OK;
}

int srt_pt_sync(srt_pt* pt) {
  int st = need_device(pt, "srt_pt_sync");
  if (st != SRT_OK) return st;
  SRT_HIP(hipStreamSynchronize(pt->stream));
  return SRT_OK;
}

}  // extern "C"
