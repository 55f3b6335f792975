// Host-side scene assembly for the path tracer and the flattened layout the kernels read.
//
// build_scene of the reference (rays/pathtracer.cpp:66-176) creates, per scene object, an
// Object{trans, itrans = trans.inverse(), has_trans, material, Tri_Mesh | Shape}, builds a BVH<Triangle>
// (leaf size 4) per mesh and one BVH<Object> (leaf size 1) over all objects.  FlatScene is the same
// information laid out for the GPU:
//   nodes      32-byte nodes, TLAS first, then every BLAS; children of an interior node are adjacent
//              (the reference allocates them back to back, student/bvh.inl:144-145)
//   tris       48 B per triangle {p0, e1 = p1 - p0, e2 = p2 - p0}, stored in BVH primitive order so a leaf
//              is a contiguous run; e1/e2 are the same fp32 subtractions Triangle::hit performs per test
//   tri_nrm    48 B per triangle {n0, n1, n2}, read only for the winning triangle of an indirect/camera ray
//   objects    one record per object in BVH<Object> primitive order: transform pair + BLAS range
//   lights     the area-light copies (List<Object> of Tri_Mesh(.., use_bvh = false)) with the matrices
//              Object::pdf composes (T = I*trans, iT = itrans*I) and, per triangle, the sample corners and
//              the 2/|cross| factor of Triangle::pdf
#ifndef SRT_PT_SCENE_H
#define SRT_PT_SCENE_H

#include <cstdint>
#include <string>
#include <vector>

namespace srt {

constexpr uint32_t LEAF_BIT = 0x80000000u;

struct Mat4 { float c[4][4]; };  // c[col][row], Mat4::cols of the reference

struct Node {
  float mn[3];
  uint32_t left;   // interior: index of the left child (right = left + 1); leaf: first primitive
  float mx[3];
  uint32_t count;  // leaf: LEAF_BIT | number of primitives; interior: 0
};
static_assert(sizeof(Node) == 32, "node layout");

struct Tri { float p0[4], e1[4], e2[4]; };     // xyz + pad
struct TriNrm { float n0[4], n1[4], n2[4]; };
static_assert(sizeof(Tri) == 48 && sizeof(TriNrm) == 48, "triangle layout");

enum : uint32_t { OBJ_MESH = 0, OBJ_SPHERE = 1 };

struct Object {
  uint32_t kind, has_trans;
  int32_t material;
  uint32_t use_bvh;            // bit 0: the mesh was built with a BVH<Triangle>; bits 8..: ordinal among the meshes with nrec > 0
  uint32_t node_base, nnodes;  // BLAS nodes [node_base, node_base + nnodes)
  uint32_t tri_base, ntri;     // triangles [tri_base, tri_base + ntri)
  float radius;
  uint32_t id;                 // 1-based insertion index (diagnostics)
  uint32_t rec_base, nrec;     // BLAS interior records [rec_base, rec_base + nrec) in blas_recs; nrec == 0: the root is a leaf
  Mat4 trans, itrans;
};
static_assert(sizeof(Object) == 48 + 128, "object layout");

struct LightTri {
  float v0[4], v1[4], v2[4];   // object-space corners (Samplers::Triangle)
  float area_term;             // 2 / |cross(T*v1 - T*v0, T*v2 - T*v0)|  (Triangle::pdf)
  float pad[3];
};

struct Light {
  uint32_t has_trans, tri_base, ntri, pad;  // tri_base indexes tris / tri_nrm / light_tris alike
  Mat4 trans, itrans;  // Object::sample
  Mat4 pdfT, pdfiT;    // Object::pdf: T = I * trans, iT = itrans * I
};

// Delta_Light (rays/light.h:57-96): Directional / Point / Spot light with the object's pose.
enum : uint32_t { DL_DIRECTIONAL = 0, DL_POINT = 1, DL_SPOT = 2 };
struct DeltaLight {
  uint32_t type, has_trans;
  float radiance[3];
  float angle_bounds[2];     // spot only (degrees, Spot_Light::angle_bounds)
  float pad;
  Mat4 trans, itrans;
};
static_assert(sizeof(DeltaLight) == 32 + 128, "delta light layout");

struct Material {
  uint32_t type;
  float a[3], b[3];
  float ior;
};

// Top-level tree in the form the wave-uniform kernel sweeps: one record per INTERIOR node of the BVH<Object>,
// in node-index order (parents before children).  A child is either another interior node (ref = its rank in
// this array) or a leaf (ref = ~first object slot, cnt = number of objects in the leaf, 0 or 1).
struct WaveInterior {
  float boxl[6], boxr[6];   // mn.xyz, mx.xyz of the left / right child
  int32_t l_ref, r_ref;
  uint32_t l_cnt, r_cnt;
};
static_assert(sizeof(WaveInterior) == 64, "wave interior layout");

struct Camera {
  Mat4 iview;
  float vert_fov, aspect_ratio;
  float screen_h, screen_w;  // tanf(Radians(fov) / 2) * 1 * 2 and ar * that (student/camera.cpp:17-18), host libm
};

struct FlatScene {
  bool use_bvh = true;
  std::vector<Node> nodes;        // [0, tlas_nodes) = BVH<Object>
  uint32_t tlas_nodes = 0;
  std::vector<Tri> tris;
  std::vector<TriNrm> tri_nrm;
  // The same records without their padding (nine floats: p0, e1, e2): what the streamed forms' cast kernel reads (pt_stream.h) - a big
  // mesh's triangles are most of what its walks miss their XCD's L2 with, and a leaf's <= 4 triangles are 144 bytes this way, 192 padded.
  std::vector<float> tri_packed;
  std::vector<Object> objects;    // BVH<Object> primitive order (insertion order when !use_bvh)
  std::vector<Light> lights;
  std::vector<LightTri> light_tris;  // indexed by (global triangle index - first light triangle)
  uint32_t light_tri_first = 0;
  std::vector<Material> materials;
  std::vector<DeltaLight> delta_lights;   // Pathtracer::point_lights, in insertion order
  uint32_t max_tlas_depth = 0, max_blas_depth = 0;  // interior-node nesting (stack frames needed)
  std::vector<WaveInterior> wave_tlas;              // empty when the root is a leaf (or in list mode)
  // Per-mesh BVH<Triangle> as interior records (both child boxes in one 64-byte fetch).  Child refs: >= 0 interior
  // rank inside the mesh's range; < 0 leaf, ~ref = (first triangle slot << 3) | triangle count (<= 4).
  std::vector<WaveInterior> blas_recs;
  std::vector<uint32_t> lazy_objects;               // object slots whose mesh has a real BVH<Triangle> (nrec > 0), by ordinal
  std::vector<uint32_t> wave_lazy;                  // per wave_tlas record: bit 0 / 1 = the left / right child's subtree holds such a mesh
};

// Input side (what the C ABI collects between scene_begin and scene_commit).
struct MeshInput {
  std::vector<float> pos, nrm;  // 3 per vertex
  std::vector<uint32_t> idx;    // 3 per triangle
};
struct ObjectInput {
  uint32_t kind = OBJ_MESH;
  Mat4 trans;
  uint32_t material = 0;
  bool is_light = false;
  float radius = 0.f;
  MeshInput mesh;
};

// Host BVH kept for srt_pt_dump_bvh (reference Node layout: bbox + start,size,l,r).
struct HostNode { float mn[3], mx[3]; uint32_t start, size, l, r; };
struct HostBVH { std::vector<HostNode> nodes; std::vector<uint32_t> prim; };

struct BuiltScene {
  FlatScene flat;
  HostBVH tlas;                 // prim = object insertion indices
  std::vector<HostBVH> blas;    // per object in insertion order (empty for spheres / list mode)
  std::vector<ObjectInput> inputs;
};

// BVH<Primitive>::build for large primitive sets on the device (pt_bvh_device.hip; same arrays as the host build, bit for
// bit).  The hook keeps pt_scene.cpp free of HIP: build_scene uses `fn` for sets of at least `min_prims` primitives when one
// is installed (thread-local: a context installs its choice around its own build_scene call).
typedef bool (*DeviceBvhBuilder)(const float* boxes6, uint32_t n, uint32_t max_leaf, HostBVH* out);
void set_device_bvh_builder(DeviceBvhBuilder fn, uint32_t min_prims);
bool build_bvh_device(const float* boxes6, uint32_t n, uint32_t max_leaf, HostBVH* out);

Mat4 mat_identity();
Mat4 mat_inverse(const Mat4& m);   // Mat4::inverse, same term order (lib/mat4.h:296-343)
Mat4 mat_mul(const Mat4& self, const Mat4& m);  // self * m  (lib/mat4.h:110-121)
bool mat_ne_identity(const Mat4& m);

// Returns "" on success, otherwise an error message (e.g. the reference's non-terminating BVH build).
std::string build_scene(const std::vector<ObjectInput>& objects, const std::vector<Material>& materials, bool use_bvh,
                        BuiltScene* out);

Camera make_camera(const float iview[16], float vert_fov_deg, float aspect_ratio);

// Delta_Light ctor: itrans = T.inverse(), has_trans = T != I.
DeltaLight make_delta_light(uint32_t type, const float radiance[3], const float angle_bounds[2], const Mat4& trans);

}  // namespace srt

#endif
