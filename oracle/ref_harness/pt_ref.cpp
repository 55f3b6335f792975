// pt_ref.cpp — harness around the REFERENCE's own path tracer (Assignments/Scotty3D/src), compiled from
// the sources where they lie under /root/reference (oracle/Makefile, target `ref`).  Nothing of the
// reference is copied into this repository.
//
// TEST INFRASTRUCTURE ONLY.  Output: oracle/_ref/libref_pt.so (git-ignored).  Used by
// tests/golden/make_pt_golden.py to produce fixtures and by tests/test_pt_oracle.py to pin
// oracle/pt_oracle.c when /root/reference is present.
//
// What runs here: PT::Pathtracer::trace_pixel and everything below it (trace, sample_direct_lighting,
// sample_indirect_lighting, BVH<>::build/hit, Triangle::hit, Sphere::hit, BBox::hit, BSDF_*::scatter,
// Samplers::*, Camera::generate_ray) exactly as the reference wrote them.  The harness supplies:
//   * the scene, fed through the reference's own constructors (Tri_Mesh, Shape, Object, BVH<Object>,
//     BSDF_*) instead of assimp (the full Scotty3D binary needs SDL2/GTK, absent in this image);
//   * the RNG seam: util/rand.cpp seeds a thread_local mt19937 from random_device/time and is
//     therefore non-deterministic; "a fixed RNG seed" (BASELINE.json north_star) needs a replacement.
//     RNG::unit/integer/coin_flip below are the SRT-RNG v1 counter-keyed generator that the oracle
//     and the HIP kernel implement too, re-keyed per (pixel, sample);
//   * a sink for Gui::Widget_Render::log_ray (the GUI's ray visualiser) that records what the path tracer hands it
//     (ref_pt_epoch_rows_log), so that the oracle's and the kernels' ray log is pinned to the reference's calls.
// -fno-access-control is applied to this translation unit only (oracle/Makefile).
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <queue>
#include <set>
#include <sstream>
#include <stack>
#include <string>
#include <thread>
#include <unordered_map>
#include <variant>
#include <vector>

// compiled with -fno-access-control (this translation unit only) to reach private members
#include "rays/pathtracer.h"
#include "scene/particles.h"
#include "rays/samplers.h"
#include "util/rand.h"
#include "gui/widgets.h"

// ------------------------------------------------------------------------------------------------
// SRT-RNG v1: PCG32 (XSH-RR 64/32) whose 64-bit state is the splitmix64 finalizer of
// (seed, pixel, sample) and whose stream constant is the (pixel, sample) key.
// ------------------------------------------------------------------------------------------------
namespace {
struct SrtRng {
  uint64_t state = 0, inc = 1;
  uint32_t draws = 0;
  void key(uint64_t seed, uint32_t pixel, uint32_t sample) {
    const uint64_t k = ((uint64_t)pixel << 32) | sample;
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (k + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    inc = (k << 1) | 1;
    state = z * 6364136223846793005ull + inc;
    draws = 0;
  }
  uint32_t next() {
    const uint64_t old = state;
    state = old * 6364136223846793005ull + inc;
    const uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
    const uint32_t rot = (uint32_t)(old >> 59);
    draws++;
    return (xs >> rot) | (xs << ((32 - rot) & 31));
  }
};
thread_local SrtRng g_rng;
}  // namespace

namespace RNG {
float unit() { return (float)(g_rng.next() >> 8) * (1.0f / 16777216.0f); }
int integer(int min, int max) {
  return min + (int)(((uint64_t)g_rng.next() * (uint64_t)(uint32_t)(max - min)) >> 32);
}
bool coin_flip(float p) { return unit() < p; }
void seed() {}
}  // namespace RNG

// The GUI's ray log; the path tracer calls it with probability 0.0005 per shading point (student/pathtracer.cpp:148).  The harness
// records the call's arguments - and which (pixel, sample) was being traced, in call order - when a log is attached.
namespace {
struct RayLog { float* buf; size_t cap, n; uint32_t pixel, sample, ordinal; };
thread_local RayLog* g_log = nullptr;
}  // namespace
void Gui::Widget_Render::log_ray(const Ray& ray, float t, Spectrum color) {
  RayLog* L = g_log;
  if (!L) return;
  if (L->n < L->cap) {
    float* e = L->buf + 13 * L->n;
    e[0] = ray.point.x; e[1] = ray.point.y; e[2] = ray.point.z;
    e[3] = ray.dir.x; e[4] = ray.dir.y; e[5] = ray.dir.z;
    e[6] = t;
    e[7] = (float)L->pixel; e[8] = (float)L->sample; e[9] = (float)L->ordinal;
    e[10] = color.r; e[11] = color.g; e[12] = color.b;
  }
  L->n++;
  L->ordinal++;
}

namespace {

struct RefPT {
  PT::Pathtracer* pt = nullptr;
  std::vector<PT::Object> objs;
  std::vector<PT::Object> lights;
  bool use_bvh = true;
};

Mat4 mat_from(const float m[16]) {  // column-major, Mat4::data order
  Mat4 r;
  for (int i = 0; i < 16; i++) r.data[i] = m[i];
  return r;
}

GL::Mesh make_mesh(const float* pos, const float* nrm, uint32_t nv, const uint32_t* idx, uint32_t ni) {
  std::vector<GL::Mesh::Vert> verts(nv);
  for (uint32_t i = 0; i < nv; i++) {
    verts[i].pos = Vec3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
    verts[i].norm = Vec3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
    verts[i].id = 0;
  }
  std::vector<GL::Mesh::Index> indices(idx, idx + ni);
  return GL::Mesh(std::move(verts), std::move(indices));
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

// Which overloads the reference's unqualified math calls resolve to matters for rounding
// (S/src/student/shapes.cpp:34-35 `sqrt(delta)`, bsdf.cpp:50 `sqrt(cos_t_squared)`): report it.
int ref_pt_unqualified_sqrt_is_double(void) { return sizeof(decltype(sqrt(1.0f))) == sizeof(double); }
int ref_pt_unqualified_pow_is_double(void) { return sizeof(decltype(pow(1.0f, 2))) == sizeof(double); }

void* ref_pt_create(uint32_t w, uint32_t h, uint32_t max_depth, int use_bvh) {
  RefPT* r = new RefPT();
  // The Widget_Render reference is only ever used by log_ray (defined above, ignores `this`).
  alignas(16) static char fake_gui[1];
  r->pt = new PT::Pathtracer(*reinterpret_cast<Gui::Widget_Render*>(fake_gui), Vec2((float)w, (float)h));
  r->pt->set_params(w, h, 1, max_depth, use_bvh != 0);
  r->use_bvh = use_bvh != 0;
  return r;
}

// type: 0 lambertian(albedo=a) 1 mirror(reflectance=a) 2 glass(transmittance=a, reflectance=b, ior)
//       3 diffuse_light(radiance=a) 4 refract(transmittance=a, ior)
// The spectra are the BSDF constructors' arguments (build_scene, rays/pathtracer.cpp:93-117).
int ref_pt_add_material(void* h, int type, const float a[3], const float b[3], float ior) {
  RefPT* r = (RefPT*)h;
  Spectrum A(a[0], a[1], a[2]), B(b[0], b[1], b[2]);
  switch (type) {
    case 0: r->pt->materials.push_back(PT::BSDF(PT::BSDF_Lambertian(A))); break;
    case 1: r->pt->materials.push_back(PT::BSDF(PT::BSDF_Mirror(A))); break;
    case 2: r->pt->materials.push_back(PT::BSDF(PT::BSDF_Glass(A, B, ior))); break;
    case 3: r->pt->materials.push_back(PT::BSDF(PT::BSDF_Diffuse(A))); break;
    case 4: r->pt->materials.push_back(PT::BSDF(PT::BSDF_Refract(A, ior))); break;
    default: return -1;
  }
  return (int)r->pt->materials.size() - 1;
}

// Object(Tri_Mesh(mesh, use_bvh), id, material, T); is_light additionally appends the
// Tri_Mesh(mesh, false) copy to area_lights (rays/pathtracer.cpp:105-116).
int ref_pt_add_mesh(void* h, const float* pos, const float* nrm, uint32_t nv, const uint32_t* idx, uint32_t ni,
                    const float T[16], uint32_t material, int is_light) {
  RefPT* r = (RefPT*)h;
  Mat4 M = mat_from(T);
  uint32_t id = (uint32_t)r->objs.size() + 1;
  if (is_light) {
    GL::Mesh lm = make_mesh(pos, nrm, nv, idx, ni);
    r->lights.push_back(PT::Object(PT::Tri_Mesh(lm, false), id, material, M));
  }
  GL::Mesh m = make_mesh(pos, nrm, nv, idx, ni);
  r->objs.emplace_back(PT::Tri_Mesh(m, r->use_bvh), id, material, M);
  return 0;
}

// An emissive Shape as build_scene handles it (rays/pathtracer.cpp:105-131).
int ref_pt_add_sphere_light(void* h, float radius, const float T[16], uint32_t material, const float* pos, const float* nrm,
                            uint32_t nv, const uint32_t* idx, uint32_t ni) {
  RefPT* r = (RefPT*)h;
  Mat4 M = mat_from(T);
  uint32_t id = (uint32_t)r->objs.size() + 1;
  GL::Mesh lm = make_mesh(pos, nrm, nv, idx, ni);
  r->lights.push_back(PT::Object(PT::Tri_Mesh(lm, false), id, material, M));
  PT::Shape shape{PT::Sphere(radius)};
  r->objs.emplace_back(std::move(shape), id, material, M);
  return 0;
}

int ref_pt_add_sphere(void* h, float radius, const float T[16], uint32_t material) {
  RefPT* r = (RefPT*)h;
  uint32_t id = (uint32_t)r->objs.size() + 1;
  PT::Shape shape{PT::Sphere(radius)};
  r->objs.emplace_back(std::move(shape), id, material, mat_from(T));
  return 0;
}

// build_lights (rays/pathtracer.cpp:26-64) for one Scene_Light: type 0 directional, 1 point, 2 spot.
int ref_pt_add_light(void* h, uint32_t type, const float radiance[3], const float angle_bounds[2], const float T[16]) {
  RefPT* r = (RefPT*)h;
  const Spectrum rad(radiance[0], radiance[1], radiance[2]);
  const Scene_ID id = 1000 + (Scene_ID)r->pt->point_lights.size();
  if (type == 0) r->pt->point_lights.push_back(PT::Delta_Light(PT::Directional_Light(rad), id, mat_from(T)));
  else if (type == 1) r->pt->point_lights.push_back(PT::Delta_Light(PT::Point_Light(rad), id, mat_from(T)));
  else if (type == 2)
    r->pt->point_lights.push_back(PT::Delta_Light(PT::Spot_Light(rad, Vec2(angle_bounds[0], angle_bounds[1])), id, mat_from(T)));
  else return -1;
  return 0;
}

// build_lights for an environment light: type 1 Env_Sphere, 2 Env_Hemisphere (uniform radiance), 0 none.
int ref_pt_set_env_light(void* h, uint32_t type, const float radiance[3]) {
  RefPT* r = (RefPT*)h;
  r->pt->env_light.reset();
  if (type == 0) return 0;
  const Spectrum rad(radiance[0], radiance[1], radiance[2]);
  if (type == 1) r->pt->env_light = PT::Env_Light(PT::Env_Sphere(rad));
  else if (type == 2) r->pt->env_light = PT::Env_Light(PT::Env_Hemisphere(rad));
  else return -1;
  return 0;
}

// Env_Map from raw pixels (HDR_Image::at(x, y) = pixels[y * w + x]).
int ref_pt_set_env_map(void* h, uint32_t w, uint32_t hh, const float* rgb) {
  RefPT* r = (RefPT*)h;
  HDR_Image img(w, hh);
  for (size_t i = 0; i < (size_t)w * hh; i++) img.at(i) = Spectrum(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
  r->pt->env_light = PT::Env_Light(PT::Env_Map(std::move(img)));
  return 0;
}

// Tail of build_scene (rays/pathtracer.cpp:165-175).
int ref_pt_commit(void* h) {
  RefPT* r = (RefPT*)h;
  r->pt->area_lights = PT::List<PT::Object>(std::move(r->lights));
  if (r->use_bvh) {
    PT::BVH<PT::Object> bvh(std::move(r->objs));
    r->pt->scene = PT::Object(std::move(bvh));
  } else {
    PT::List<PT::Object> list(std::move(r->objs));
    r->pt->scene = PT::Object(std::move(list));
  }
  r->objs.clear();
  r->lights.clear();
  return 0;
}

int ref_pt_set_camera(void* h, const float iview[16], float vert_fov_deg, float aspect_ratio) {
  RefPT* r = (RefPT*)h;
  Camera& c = r->pt->camera;
  c.iview = mat_from(iview);
  c.view = c.iview.inverse();
  c.vert_fov = vert_fov_deg;
  c.aspect_ratio = aspect_ratio;
  return 0;
}

// The reference camera's own iview for a look_at pose (util/camera.cpp:31-40,141-146).
int ref_pt_lookat_iview(const float pos[3], const float center[3], float iview_out[16]) {
  Camera c(Vec2(1.0f, 1.0f));
  c.look_at(Vec3(center[0], center[1], center[2]), Vec3(pos[0], pos[1], pos[2]));
  for (int i = 0; i < 16; i++) iview_out[i] = c.iview.data[i];
  return 0;
}

// trace_pixel for a list of (x, y, sample) with the RNG re-keyed before every call.
// rgb_out: 3 floats per sample (emissive + reflected, student/pathtracer.cpp:14-40); draws_out (nullable):
// RNG draws consumed by that sample.
int ref_pt_trace_samples(void* h, uint64_t seed, const uint32_t* xs, const uint32_t* ys, const uint32_t* ss,
                         size_t n, float* rgb_out, uint32_t* draws_out) {
  RefPT* r = (RefPT*)h;
  const uint32_t w = (uint32_t)r->pt->out_w;
  for (size_t k = 0; k < n; k++) {
    g_rng.key(seed, ys[k] * w + xs[k], ss[k]);
    Spectrum p = r->pt->trace_pixel(xs[k], ys[k]);
    rgb_out[3 * k] = p.r; rgb_out[3 * k + 1] = p.g; rgb_out[3 * k + 2] = p.b;
    if (draws_out) draws_out[k] = g_rng.draws;
  }
  return 0;
}

// One epoch as do_trace computes it (rays/pathtracer.cpp:209-231): per pixel, the mean of the valid
// samples sample_base .. sample_base+samples-1.  img_out: w*h*3 floats, row 0 = bottom (HDR_Image::at).
int ref_pt_epoch(void* h, uint64_t seed, uint32_t sample_base, uint32_t samples, float* img_out) {
  RefPT* r = (RefPT*)h;
  const size_t w = r->pt->out_w, hh = r->pt->out_h;
  for (size_t j = 0; j < hh; j++) {
    for (size_t i = 0; i < w; i++) {
      Spectrum acc;
      size_t sampled = 0;
      for (uint32_t s = 0; s < samples; s++) {
        g_rng.key(seed, (uint32_t)(j * w + i), sample_base + s);
        Spectrum p = r->pt->trace_pixel(i, j);
        if (p.valid()) { acc += p; sampled++; }
      }
      if (sampled > 0) acc *= (1.0f / sampled);
      float* o = img_out + 3 * (j * w + i);
      o[0] = acc.r; o[1] = acc.g; o[2] = acc.b;
    }
  }
  return 0;
}

// The same for rows [row0, row1) only (img_out still addresses the whole image): lets a caller spread one epoch
// over host threads the way the reference's thread pool spreads epochs.  trace_pixel only reads the scene and
// the RNG state is thread_local, so concurrent calls on one handle are safe.
int ref_pt_epoch_rows(void* h, uint64_t seed, uint32_t sample_base, uint32_t samples, uint32_t row0, uint32_t row1,
                      float* img_out) {
  RefPT* r = (RefPT*)h;
  const size_t w = r->pt->out_w, hh = r->pt->out_h;
  for (size_t j = row0; j < row1 && j < hh; j++) {
    for (size_t i = 0; i < w; i++) {
      Spectrum acc;
      size_t sampled = 0;
      for (uint32_t s = 0; s < samples; s++) {
        g_rng.key(seed, (uint32_t)(j * w + i), sample_base + s);
        Spectrum p = r->pt->trace_pixel(i, j);
        if (p.valid()) { acc += p; sampled++; }
      }
      if (sampled > 0) acc *= (1.0f / sampled);
      float* o = img_out + 3 * (j * w + i);
      o[0] = acc.r; o[1] = acc.g; o[2] = acc.b;
    }
  }
  return 0;
}

// ref_pt_epoch_rows with the ray log attached: log13 = 13 floats per log_ray call {ray.point, ray.dir, t, pixel, sample, ordinal of
// the call within its sample, color}; *n_logged counts every call, also those beyond cap.
int ref_pt_epoch_rows_log(void* h, uint64_t seed, uint32_t sample_base, uint32_t samples, uint32_t row0, uint32_t row1, float* img_out,
                          float* log13, size_t cap, size_t* n_logged) {
  RefPT* r = (RefPT*)h;
  const size_t w = r->pt->out_w, hh = r->pt->out_h;
  RayLog L{log13, log13 ? cap : 0, 0, 0, 0, 0};
  g_log = &L;
  for (size_t j = row0; j < row1 && j < hh; j++) {
    for (size_t i = 0; i < w; i++) {
      Spectrum acc;
      size_t sampled = 0;
      for (uint32_t s = 0; s < samples; s++) {
        g_rng.key(seed, (uint32_t)(j * w + i), sample_base + s);
        L.pixel = (uint32_t)(j * w + i); L.sample = sample_base + s; L.ordinal = 0;
        Spectrum p = r->pt->trace_pixel(i, j);
        if (p.valid()) { acc += p; sampled++; }
      }
      if (sampled > 0) acc *= (1.0f / sampled);
      if (img_out) { float* o = img_out + 3 * (j * w + i); o[0] = acc.r; o[1] = acc.g; o[2] = acc.b; }
    }
  }
  g_log = nullptr;
  if (n_logged) *n_logged = L.n;
  return 0;
}

// HDR_Image::tonemap_to on caller-supplied radiance.  The reference's loop reads the image's `exposure` MEMBER (what the
// display path's get_texture(e) leaves behind), not its argument: the harness sets both to the same value.
int ref_pt_tonemap(uint32_t w, uint32_t h, const float* rgb, float exposure, unsigned char* rgba) {
  HDR_Image img(w, h);
  for (size_t i = 0; i < (size_t)w * h; i++) img.at(i) = Spectrum(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
  img.exposure = exposure;
  std::vector<unsigned char> data;
  img.tonemap_to(data, exposure);
  memcpy(rgba, data.data(), data.size());
  return 0;
}

// scene.hit for explicit rays: hit flag, distance, position, normal, material (Trace, rays/trace.h).
int ref_pt_hit(void* h, const float* org, const float* dir, const float* bounds, size_t n, float* out8) {
  RefPT* r = (RefPT*)h;
  for (size_t k = 0; k < n; k++) {
    Ray ray;
    ray.point = Vec3(org[3 * k], org[3 * k + 1], org[3 * k + 2]);
    ray.dir = Vec3(dir[3 * k], dir[3 * k + 1], dir[3 * k + 2]);
    ray.dist_bounds = Vec2(bounds[2 * k], bounds[2 * k + 1]);
    PT::Trace t = r->pt->scene.hit(ray);
    float* o = out8 + 9 * k;
    o[0] = t.hit ? 1.0f : 0.0f; o[1] = t.distance;
    o[2] = t.position.x; o[3] = t.position.y; o[4] = t.position.z;
    o[5] = t.normal.x; o[6] = t.normal.y; o[7] = t.normal.z;
    o[8] = (float)t.material;
  }
  return 0;
}

// Scene_Particles::Particle::update (student/particles.cpp:5-59) for n particles against this scene - the loop body of
// Scene_Particles::step2 (scene/particles.cpp:134-138).  pos / vel: 3 floats per particle, age: 1; alive[k] = update()'s verdict.
int ref_pt_particles_update(void* h, float* pos, float* vel, float* age, size_t n, float dt, float radius, unsigned char* alive) {
  RefPT* r = (RefPT*)h;
  for (size_t k = 0; k < n; k++) {
    Scene_Particles::Particle p;
    p.pos = Vec3(pos[3 * k], pos[3 * k + 1], pos[3 * k + 2]);
    p.velocity = Vec3(vel[3 * k], vel[3 * k + 1], vel[3 * k + 2]);
    p.age = age[k];
    alive[k] = p.update(r->pt->scene, dt, radius) ? 1 : 0;
    pos[3 * k] = p.pos.x; pos[3 * k + 1] = p.pos.y; pos[3 * k + 2] = p.pos.z;
    vel[3 * k] = p.velocity.x; vel[3 * k + 1] = p.velocity.y; vel[3 * k + 2] = p.velocity.z;
    age[k] = p.age;
  }
  return 0;
}

// Dump a BVH's node array: per node 6 floats bbox + 4 u32 (start,size,l,r).  which = -1: the scene's
// BVH<Object>; which >= 0: the BVH<Triangle> of the which-th primitive of the scene BVH (in the
// BVH's own, post-build primitive order).  Returns the node count (writes at most cap nodes).
long ref_pt_dump_bvh(void* h, int which, float* boxes, uint32_t* links, size_t cap, uint32_t* prim_order) {
  RefPT* r = (RefPT*)h;
  auto* sb = std::get_if<PT::BVH<PT::Object>>(&r->pt->scene.underlying);
  if (!sb) return -1;
  if (which < 0) {
    size_t n = sb->nodes.size();
    for (size_t i = 0; i < n && i < cap; i++) {
      const auto& nd = sb->nodes[i];
      float* b = boxes + 6 * i;
      b[0] = nd.bbox.min.x; b[1] = nd.bbox.min.y; b[2] = nd.bbox.min.z;
      b[3] = nd.bbox.max.x; b[4] = nd.bbox.max.y; b[5] = nd.bbox.max.z;
      uint32_t* l = links + 4 * i;
      l[0] = (uint32_t)nd.start; l[1] = (uint32_t)nd.size; l[2] = (uint32_t)nd.l; l[3] = (uint32_t)nd.r;
    }
    if (prim_order)
      for (size_t i = 0; i < sb->primitives.size(); i++) prim_order[i] = sb->primitives[i]._id;
    return (long)n;
  }
  if ((size_t)which >= sb->primitives.size()) return -1;
  auto* tm = std::get_if<PT::Tri_Mesh>(&sb->primitives[which].underlying);
  if (!tm) return -2;
  const auto& tb = tm->triangle_bvh;
  size_t n = tb.nodes.size();
  for (size_t i = 0; i < n && i < cap; i++) {
    const auto& nd = tb.nodes[i];
    float* b = boxes + 6 * i;
    b[0] = nd.bbox.min.x; b[1] = nd.bbox.min.y; b[2] = nd.bbox.min.z;
    b[3] = nd.bbox.max.x; b[4] = nd.bbox.max.y; b[5] = nd.bbox.max.z;
    uint32_t* l = links + 4 * i;
    l[0] = (uint32_t)nd.start; l[1] = (uint32_t)nd.size; l[2] = (uint32_t)nd.l; l[3] = (uint32_t)nd.r;
  }
  if (prim_order)
    for (size_t i = 0; i < tb.primitives.size(); i++) prim_order[i] = tb.primitives[i].v0;  // first vertex index
  return (long)n;
}

}  // extern "C"
#pragma GCC visibility pop
