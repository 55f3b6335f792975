// Host-side build_scene for the MI355X path tracer: Object construction, BVH builds that are
// structure-identical to the reference's, and flattening into the layout of pt_scene.h.
// Reference paths are relative to /root/reference/Assignments/Scotty3D/src/.
#include "pt_scene.h"

#include <cfloat>
#include <cmath>
#include <algorithm>
#include <cstring>

namespace srt {

namespace {

struct Box {
  float mn[3], mx[3];
  Box() { for (int i = 0; i < 3; i++) { mn[i] = FLT_MAX; mx[i] = -FLT_MAX; } }  // BBox(), lib/bbox.h:17
  void enclose(const float p[3]) {
    for (int i = 0; i < 3; i++) { mn[i] = std::min(mn[i], p[i]); mx[i] = std::max(mx[i], p[i]); }
  }
  void enclose(const Box& b) {
    for (int i = 0; i < 3; i++) { mn[i] = std::min(mn[i], b.mn[i]); mx[i] = std::max(mx[i], b.mx[i]); }
  }
  float center(int axis) const { return (mn[axis] + mx[axis]) * 0.5f; }
  float surface_area() const {  // lib/bbox.h:50-54
    if (mn[0] > mx[0] || mn[1] > mx[1] || mn[2] > mx[2]) return 0.0f;
    const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    return 2.0f * (ex * ez + ex * ey + ey * ez);
  }
  void transform(const Mat4& t) {  // lib/bbox.h:57-73
    float amin[3], amax[3];
    for (int i = 0; i < 3; i++) { amin[i] = mn[i]; amax[i] = mx[i]; mn[i] = mx[i] = t.c[3][i]; }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        const float a = t.c[j][i] * amin[j], b = t.c[j][i] * amax[j];
        if (a < b) { mn[i] += a; mx[i] += b; } else { mn[i] += b; mx[i] += a; }
      }
  }
};

// Triangle::bbox (student/tri_mesh.cpp:7-30): zero-extent axes are widened by +1 on the max side.
Box triangle_box(const float* p0, const float* p1, const float* p2) {
  Box b;
  for (int a = 0; a < 3; a++) {
    const float lo = std::min({p0[a], p1[a], p2[a]});
    float hi = std::max({p0[a], p1[a], p2[a]});
    hi = (lo >= hi) ? (lo + 1.0f) : hi;
    b.mn[a] = lo; b.mx[a] = hi;
  }
  return b;
}

// std::partition as libstdc++ implements it for bidirectional iterators; the permutation it
// leaves behind decides the order of primitives inside leaves (and with it Trace::min tie-breaks).
uint32_t partition_by_center(std::vector<uint32_t>& prim, const std::vector<Box>& boxes, uint32_t first, uint32_t last,
                             int axis, float line) {
  auto pred = [&](uint32_t slot) { return boxes[prim[slot]].center(axis) < line; };
  while (true) {
    while (true) {
      if (first == last) return first;
      if (pred(first)) ++first; else break;
    }
    --last;
    while (true) {
      if (first == last) return first;
      if (!pred(last)) --last; else break;
    }
    std::swap(prim[first], prim[last]);
    ++first;
  }
}

// BVH<Primitive>::build (student/bvh.inl:35-163): level order; per axis up to nine candidate planes
// min + k*interval (float accumulation), SAH cost, axis chosen by exact float equality with the minimum.
thread_local DeviceBvhBuilder g_device_builder = nullptr;
thread_local uint32_t g_device_min = 0;

bool build_bvh(const std::vector<Box>& boxes, uint32_t max_leaf, HostBVH* out) {
  const uint32_t n = (uint32_t)boxes.size();
  static_assert(sizeof(Box) == 6 * sizeof(float), "Box is six floats");
  if (g_device_builder && n >= g_device_min && n > max_leaf && g_device_builder(&boxes[0].mn[0], n, max_leaf, out)) return true;
  // (a device build that fails - no termination, or no memory - falls through: the host build gives the verdict)
  out->nodes.clear();
  out->prim.resize(n);
  for (uint32_t i = 0; i < n; i++) out->prim[i] = i;
  auto new_node = [&](const Box& b, uint32_t start, uint32_t size) {
    HostNode nd;
    for (int i = 0; i < 3; i++) { nd.mn[i] = b.mn[i]; nd.mx[i] = b.mx[i]; }
    nd.start = start; nd.size = size; nd.l = 0; nd.r = 0;
    out->nodes.push_back(nd);
  };
  Box all;
  for (const Box& b : boxes) all.enclose(b);
  new_node(all, 0, n);
  const size_t node_limit = 8ull * n + 64;
  struct Split { Box left, right; int nl = 0, nr = 0; float line = 0; };
  for (size_t cur = 0; cur < out->nodes.size(); cur++) {
    if (out->nodes[cur].size <= max_leaf) continue;
    if (out->nodes.size() > node_limit) return false;  // the reference would never return
    const HostNode nd = out->nodes[cur];
    Box nbox;
    for (int i = 0; i < 3; i++) { nbox.mn[i] = nd.mn[i]; nbox.mx[i] = nd.mx[i]; }
    const uint32_t start = nd.start, end = nd.start + nd.size;
    float best_cost[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    Split best[3];
    for (int axis = 0; axis < 3; axis++) {
      Split best_axis;
      const float interval = (nbox.mx[axis] - nbox.mn[axis]) / (float)10;
      for (float plane = nbox.mn[axis] + interval; plane < nbox.mx[axis]; plane += interval) {
        const uint32_t mid = partition_by_center(out->prim, boxes, start, end, axis, plane);
        Split s;
        s.line = plane;
        for (uint32_t i = start; i < end; i++) {
          if (i >= mid) { s.right.enclose(boxes[out->prim[i]]); s.nr++; }
          else { s.left.enclose(boxes[out->prim[i]]); s.nl++; }
        }
        const float cost = s.left.surface_area() / nbox.surface_area() * (float)s.nl +
                           s.right.surface_area() / nbox.surface_area() * (float)s.nr + 1.0f;
        if (cost < best_cost[axis]) { best_cost[axis] = cost; best_axis = s; }
      }
      best[axis] = best_axis;
    }
    const float lowest = std::min(best_cost[0], std::min(best_cost[1], best_cost[2]));
    const int axis = (lowest == best_cost[0]) ? 0 : ((lowest == best_cost[1]) ? 1 : 2);
    const uint32_t l = (uint32_t)out->nodes.size();
    partition_by_center(out->prim, boxes, start, end, axis, best[axis].line);
    new_node(best[axis].left, nd.start, (uint32_t)best[axis].nl);
    new_node(best[axis].right, nd.start + (uint32_t)best[axis].nl, (uint32_t)best[axis].nr);
    out->nodes[cur].l = l;
    out->nodes[cur].r = l + 1;
  }
  return true;
}

}  // namespace
void set_device_bvh_builder(DeviceBvhBuilder fn, uint32_t min_prims) { g_device_builder = fn; g_device_min = min_prims; }
namespace {

// Interior-node nesting of the tree.  Children are allocated after their parent (level order, student/bvh.inl:144-145),
// so one backward pass over the node array does it - no recursion, however skewed the tree.
uint32_t interior_depth(const HostBVH& b) {
  if (b.nodes.empty()) return 0;
  std::vector<uint32_t> d(b.nodes.size(), 0u);
  for (size_t n = b.nodes.size(); n-- > 0;) {
    const HostNode& nd = b.nodes[n];
    if (nd.l != nd.r) d[n] = 1u + std::max(d[nd.l], d[nd.r]);
  }
  return d[0];
}

// Mat4 * Vec3 with perspective divide (lib/mat4.h:125-131).
void mat_point(const Mat4& m, const float v[3], float out[3]) {
  float o[4];
  for (int j = 0; j < 4; j++) o[j] = ((m.c[0][j] * v[0] + m.c[1][j] * v[1]) + m.c[2][j] * v[2]) + m.c[3][j] * 1.0f;
  out[0] = o[0] / o[3]; out[1] = o[1] / o[3]; out[2] = o[2] / o[3];
}

void append_nodes(const HostBVH& b, FlatScene* f) {
  for (const HostNode& h : b.nodes) {
    Node n;
    for (int i = 0; i < 3; i++) { n.mn[i] = h.mn[i]; n.mx[i] = h.mx[i]; }
    if (h.l == h.r) { n.left = h.start; n.count = LEAF_BIT | h.size; }
    else { n.left = h.l; n.count = 0; }
    f->nodes.push_back(n);
  }
}

// Term tables of Mat4::inverse / Mat4::det: digit pairs are (col,row); the order of terms and of the
// factors inside a term fixes the fp32 rounding, so it is data (lib/mat4.h:206-231, 296-343).
const char* const kInverseTerms[16] = {
    "+122331-132231+132132-112332-122133+112233", "+032231-022331-032132+012332+022133-012233",
    "+021331-031231+031132-011332-021133+011233", "+031221-021321-031122+011322+021123-011223",
    "+132230-122330-132032+102332+122033-102233", "+022330-032230+032032-002332-022033+002233",
    "+031230-021330-031032+001332+021033-001233", "+021320-031220+031022-001322-021023+001223",
    "+112330-132130+132031-102331-112033+102133", "+032130-012330-032031+002331+012033-002133",
    "+011330-031130+031031-001331-011033+001133", "+031120-011320-031021+001321+011023-001123",
    "+122130-112230-122031+102231+112032-102132", "+012230-022130+022031-002231-012032+002132",
    "+021130-011230-021031+001231+011032-001132", "+011220-021120+021021-001221-011022+001122"};
const char* const kDetTerms =
    "+03122130-02132130-03112230+01132230+02112330-01122330-03122031+02132031+03102231-00132231-02102331"
    "+00122331+03112032-01132032-03102132+00132132+01102332-00112332-02112033+01122033+02102133-00122133"
    "-01102233+00112233";

float signed_products(const Mat4& m, const char* t, int factors) {
  float acc = 0.0f;
  bool first = true;
  while (*t) {
    const bool minus = (*t++ == '-');
    float p = m.c[t[0] - '0'][t[1] - '0'];
    for (int f = 1; f < factors; f++) p = p * m.c[t[2 * f] - '0'][t[2 * f + 1] - '0'];
    t += 2 * factors;
    if (first) { acc = p; first = false; }
    else acc = minus ? acc - p : acc + p;
  }
  return acc;
}

}  // namespace

Mat4 mat_identity() {
  Mat4 r;
  std::memset(&r, 0, sizeof r);
  r.c[0][0] = r.c[1][1] = r.c[2][2] = r.c[3][3] = 1.0f;
  return r;
}

Mat4 mat_inverse(const Mat4& m) {
  Mat4 r;
  for (int e = 0; e < 16; e++) r.c[e / 4][e % 4] = signed_products(m, kInverseTerms[e], 3);
  const float det = signed_products(m, kDetTerms, 4);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) r.c[i][j] /= det;
  return r;
}

Mat4 mat_mul(const Mat4& self, const Mat4& m) {
  Mat4 r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float acc = 0.0f;
      for (int k = 0; k < 4; k++) acc += m.c[i][k] * self.c[k][j];
      r.c[i][j] = acc;
    }
  return r;
}

bool mat_ne_identity(const Mat4& m) {  // operator!= compares values, so -0.0f equals 0.0f
  const Mat4 id = mat_identity();
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      if (m.c[i][j] != id.c[i][j]) return true;
  return false;
}

DeltaLight make_delta_light(uint32_t type, const float radiance[3], const float angle_bounds[2], const Mat4& trans) {
  DeltaLight l;
  std::memset(&l, 0, sizeof l);
  l.type = type;
  for (int i = 0; i < 3; i++) l.radiance[i] = radiance[i];
  l.angle_bounds[0] = angle_bounds ? angle_bounds[0] : 0.0f;
  l.angle_bounds[1] = angle_bounds ? angle_bounds[1] : 0.0f;
  l.trans = trans;
  l.itrans = mat_inverse(trans);
  l.has_trans = mat_ne_identity(trans) ? 1u : 0u;
  return l;
}

Camera make_camera(const float iview[16], float vert_fov_deg, float aspect_ratio) {
  Camera c;
  std::memcpy(&c.iview, iview, sizeof(Mat4));
  c.vert_fov = vert_fov_deg;
  c.aspect_ratio = aspect_ratio;
  // student/camera.cpp:17-18; Radians(v) = v * (PI_F / 180.0f).  tanf is host libm, as in the reference.
  const float PI_F = 3.14159265358979323846264338327950288f;
  c.screen_h = std::tan((vert_fov_deg * (PI_F / 180.0f)) / 2.0f) * 1.0f * 2.0f;
  c.screen_w = aspect_ratio * c.screen_h;
  return c;
}

std::string build_scene(const std::vector<ObjectInput>& objects, const std::vector<Material>& materials, bool use_bvh,
                        BuiltScene* out) {
  BuiltScene& B = *out;
  B = BuiltScene();
  B.inputs = objects;
  FlatScene& F = B.flat;
  F.use_bvh = use_bvh;
  F.materials = materials;
  for (Material& m : F.materials)
    if (m.type == 0)  // BSDF_Lambertian(albedo) : albedo(albedo / PI_F)   rays/bsdf.h:26
      for (int i = 0; i < 3; i++) m.a[i] = m.a[i] / 3.14159265358979323846264338327950288f;

  const uint32_t nobj = (uint32_t)objects.size();
  for (uint32_t i = 0; i < nobj; i++) {
    const ObjectInput& in = objects[i];
    if (in.material >= materials.size()) return "object " + std::to_string(i) + " references an unknown material";
    if (in.kind == OBJ_MESH || (in.is_light && !in.mesh.idx.empty())) {   // a mesh, or an emissive shape's light mesh
      if (in.mesh.idx.empty() || in.mesh.idx.size() % 3) return "mesh " + std::to_string(i) + " has no triangles";
      for (uint32_t v : in.mesh.idx)
        if ((size_t)v * 3 + 2 >= in.mesh.pos.size()) return "mesh " + std::to_string(i) + " has an out-of-range vertex index";
    }
  }

  // Per-mesh BVH<Triangle> (Tri_Mesh::build, student/tri_mesh.cpp:145-170, leaf size 4) and object boxes.
  B.blas.resize(nobj);
  std::vector<Box> obj_boxes(nobj);
  std::vector<Mat4> itrans(nobj);
  std::vector<bool> has_trans(nobj);
  for (uint32_t i = 0; i < nobj; i++) {
    const ObjectInput& in = objects[i];
    itrans[i] = mat_inverse(in.trans);            // Object ctor, rays/object.h:18-21
    has_trans[i] = mat_ne_identity(in.trans);
    Box ob;
    if (in.kind == OBJ_SPHERE) {                  // Sphere::bbox, student/shapes.cpp:9-15
      const float lo[3] = {-in.radius, -in.radius, -in.radius}, hi[3] = {in.radius, in.radius, in.radius};
      ob.enclose(lo);
      ob.enclose(hi);
    } else {
      const uint32_t ntri = (uint32_t)in.mesh.idx.size() / 3;
      std::vector<Box> tb(ntri);
      for (uint32_t t = 0; t < ntri; t++)
        tb[t] = triangle_box(&in.mesh.pos[3 * in.mesh.idx[3 * t]], &in.mesh.pos[3 * in.mesh.idx[3 * t + 1]],
                             &in.mesh.pos[3 * in.mesh.idx[3 * t + 2]]);
      if (use_bvh) {
        if (!build_bvh(tb, 4, &B.blas[i]))
          return "BVH<Triangle> build of object " + std::to_string(i) +
                 " does not terminate (coincident centroids); the reference loops forever on this mesh";
        const HostNode& root = B.blas[i].nodes[0];
        for (int a = 0; a < 3; a++) { ob.mn[a] = root.mn[a]; ob.mx[a] = root.mx[a]; }
      } else {
        for (const Box& b : tb) ob.enclose(b);  // List<Triangle>::bbox
      }
    }
    if (has_trans[i]) ob.transform(in.trans);     // Object::bbox, rays/object.h:51-55
    obj_boxes[i] = ob;
  }

  // BVH<Object> (leaf size 1) or List<Object>.
  if (use_bvh) {
    if (!build_bvh(obj_boxes, 1, &B.tlas)) return "BVH<Object> build does not terminate (coincident object centroids)";
  } else {
    B.tlas.nodes.clear();
    B.tlas.prim.resize(nobj);
    for (uint32_t i = 0; i < nobj; i++) B.tlas.prim[i] = i;
  }

  // Flatten.
  if (use_bvh) append_nodes(B.tlas, &F);
  F.tlas_nodes = (uint32_t)F.nodes.size();
  F.max_tlas_depth = use_bvh ? interior_depth(B.tlas) : 0;
  if (use_bvh) {  // interior-node sweep order for the wave-uniform kernel
    const std::vector<HostNode>& N = B.tlas.nodes;
    std::vector<int32_t> rank(N.size(), -1);
    int32_t q = 0;
    for (size_t n = 0; n < N.size(); n++)
      if (N[n].l != N[n].r) rank[n] = q++;
    for (size_t n = 0; n < N.size(); n++) {
      if (N[n].l == N[n].r) continue;
      WaveInterior wi;
      const HostNode& a = N[N[n].l];
      const HostNode& b = N[N[n].r];
      for (int i = 0; i < 3; i++) { wi.boxl[i] = a.mn[i]; wi.boxl[3 + i] = a.mx[i]; wi.boxr[i] = b.mn[i]; wi.boxr[3 + i] = b.mx[i]; }
      wi.l_ref = (a.l != a.r) ? rank[N[n].l] : ~(int32_t)a.start;
      wi.r_ref = (b.l != b.r) ? rank[N[n].r] : ~(int32_t)b.start;
      wi.l_cnt = a.size;
      wi.r_cnt = b.size;
      F.wave_tlas.push_back(wi);
    }
  }
  auto append_triangles = [&](const MeshInput& m, const std::vector<uint32_t>* order) {
    const uint32_t ntri = (uint32_t)m.idx.size() / 3;
    for (uint32_t k = 0; k < ntri; k++) {
      const uint32_t t = order ? (*order)[k] : k;
      const float* p0 = &m.pos[3 * m.idx[3 * t]];
      const float* p1 = &m.pos[3 * m.idx[3 * t + 1]];
      const float* p2 = &m.pos[3 * m.idx[3 * t + 2]];
      Tri g;
      TriNrm nn;
      std::memset(&g, 0, sizeof g);
      std::memset(&nn, 0, sizeof nn);
      for (int a = 0; a < 3; a++) {
        g.p0[a] = p0[a];
        g.e1[a] = p1[a] - p0[a];  // p0p1, student/tri_mesh.cpp:60
        g.e2[a] = p2[a] - p0[a];  // p0p2
        nn.n0[a] = m.nrm[3 * m.idx[3 * t] + a];
        nn.n1[a] = m.nrm[3 * m.idx[3 * t + 1] + a];
        nn.n2[a] = m.nrm[3 * m.idx[3 * t + 2] + a];
      }
      F.tris.push_back(g);
      F.tri_nrm.push_back(nn);
      for (int a = 0; a < 3; a++) F.tri_packed.push_back(g.p0[a]);
      for (int a = 0; a < 3; a++) F.tri_packed.push_back(g.e1[a]);
      for (int a = 0; a < 3; a++) F.tri_packed.push_back(g.e2[a]);
    }
  };
  for (uint32_t slot = 0; slot < nobj; slot++) {
    const uint32_t i = B.tlas.prim[slot];
    const ObjectInput& in = objects[i];
    Object o;
    std::memset(&o, 0, sizeof o);
    o.kind = in.kind;
    o.has_trans = has_trans[i] ? 1u : 0u;
    o.material = (int32_t)in.material;
    o.use_bvh = (in.kind == OBJ_MESH && use_bvh) ? 1u : 0u;
    o.radius = in.radius;
    o.id = i + 1;
    o.trans = in.trans;
    o.itrans = itrans[i];
    if (in.kind == OBJ_MESH) {
      o.tri_base = (uint32_t)F.tris.size();
      o.ntri = (uint32_t)in.mesh.idx.size() / 3;
      if (use_bvh) {
        o.node_base = (uint32_t)F.nodes.size();
        o.nnodes = (uint32_t)B.blas[i].nodes.size();
        append_nodes(B.blas[i], &F);
        F.max_blas_depth = std::max(F.max_blas_depth, interior_depth(B.blas[i]));
        {  // interior records of this BLAS
          const std::vector<HostNode>& N = B.blas[i].nodes;
          std::vector<int32_t> rank(N.size(), -1);
          int32_t q = 0;
          for (size_t n = 0; n < N.size(); n++)
            if (N[n].l != N[n].r) rank[n] = q++;
          o.rec_base = (uint32_t)F.blas_recs.size();
          o.nrec = (uint32_t)q;
          auto ref_of = [&](uint32_t child) -> int32_t {
            const HostNode& c = N[child];
            if (c.l != c.r) return rank[child];
            return ~(int32_t)((c.start << 3) | (c.size & 7u));
          };
          for (size_t n = 0; n < N.size(); n++) {
            if (N[n].l == N[n].r) continue;
            WaveInterior wi;
            const HostNode& a = N[N[n].l];
            const HostNode& b = N[N[n].r];
            for (int k = 0; k < 3; k++) { wi.boxl[k] = a.mn[k]; wi.boxl[3 + k] = a.mx[k]; wi.boxr[k] = b.mn[k]; wi.boxr[3 + k] = b.mx[k]; }
            wi.l_ref = ref_of(N[n].l);
            wi.r_ref = ref_of(N[n].r);
            wi.l_cnt = a.size;
            wi.r_cnt = b.size;
            F.blas_recs.push_back(wi);
          }
        }
        append_triangles(in.mesh, &B.blas[i].prim);
      } else {
        append_triangles(in.mesh, nullptr);
      }
    }
    if (o.kind == OBJ_MESH && o.use_bvh != 0u && o.nrec > 0u) {
      // a mesh with a real BVH<Triangle>: its ordinal among those (the streamed sweeps queue its walks per ordinal)
      o.use_bvh |= (uint32_t)F.lazy_objects.size() << 8;
      F.lazy_objects.push_back(slot);
    }
    F.objects.push_back(o);
  }

  // Which children of the top-level records hold a mesh with a real BVH<Triangle> somewhere below (children come after
  // their parent in sweep order, so one backward pass does it).
  F.wave_lazy.assign(F.wave_tlas.size(), 0u);
  for (size_t q = F.wave_tlas.size(); q-- > 0;) {
    const WaveInterior& w = F.wave_tlas[q];
    auto child_has = [&](int32_t ref, uint32_t cnt) {
      if (ref >= 0) return F.wave_lazy[(size_t)ref] != 0u;
      const uint32_t first = (uint32_t)~ref;
      for (uint32_t k = first; k < first + cnt && k < F.objects.size(); k++)
        if (F.objects[k].kind == OBJ_MESH && F.objects[k].use_bvh != 0u && F.objects[k].nrec > 0u) return true;
      return false;
    };
    F.wave_lazy[q] = (child_has(w.l_ref, w.l_cnt) ? 1u : 0u) | (child_has(w.r_ref, w.r_cnt) ? 2u : 0u);
  }

  // Area lights: List<Object> of Tri_Mesh(mesh, false) in insertion order (rays/pathtracer.cpp:105-116,163).
  F.light_tri_first = (uint32_t)F.tris.size();
  for (uint32_t i = 0; i < nobj; i++) {
    const ObjectInput& in = objects[i];
    // an emissive analytic shape is intersected as the shape but lit through its triangle approximation
    // (obj.posed_mesh(), rays/pathtracer.cpp:105-116): its ObjectInput carries that mesh next to the radius
    if (!in.is_light || in.mesh.idx.empty()) continue;
    Light L;
    std::memset(&L, 0, sizeof L);
    L.has_trans = has_trans[i] ? 1u : 0u;
    L.tri_base = (uint32_t)F.tris.size();
    L.ntri = (uint32_t)in.mesh.idx.size() / 3;
    L.trans = in.trans;
    L.itrans = itrans[i];
    const Mat4 id = mat_identity();
    L.pdfT = id;
    L.pdfiT = id;
    if (has_trans[i]) {  // Object::pdf, rays/object.h:90-94
      L.pdfT = mat_mul(id, in.trans);
      L.pdfiT = mat_mul(itrans[i], id);
    }
    append_triangles(in.mesh, nullptr);
    for (uint32_t t = 0; t < L.ntri; t++) {
      LightTri lt;
      std::memset(&lt, 0, sizeof lt);
      const float* v[3] = {&in.mesh.pos[3 * in.mesh.idx[3 * t]], &in.mesh.pos[3 * in.mesh.idx[3 * t + 1]],
                           &in.mesh.pos[3 * in.mesh.idx[3 * t + 2]]};
      float w[3][3];
      for (int k = 0; k < 3; k++) mat_point(L.pdfT, v[k], w[k]);
      for (int a = 0; a < 3; a++) { lt.v0[a] = v[0][a]; lt.v1[a] = v[1][a]; lt.v2[a] = v[2][a]; }
      // a = 2.0f / cross(v_1 - v_0, v_2 - v_0).norm()   (student/tri_mesh.cpp:137)
      const float ax = w[1][0] - w[0][0], ay = w[1][1] - w[0][1], az = w[1][2] - w[0][2];
      const float bx = w[2][0] - w[0][0], by = w[2][1] - w[0][1], bz = w[2][2] - w[0][2];
      const float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
      lt.area_term = 2.0f / std::sqrt(cx * cx + cy * cy + cz * cz);
      F.light_tris.push_back(lt);
    }
    F.lights.push_back(L);
  }
  return "";
}

}  // namespace srt
