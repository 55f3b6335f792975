// Wave-uniform persistent path-tracing kernel for scenes with a small BVH<Object> (<= kWaveMaxObjects
// objects — the Cornell-box family of BASELINE configs 3-5).
//
// Why a second kernel.  The reference's traversal (student/bvh.inl:166-223) visits almost the whole
// top-level tree for every ray: a far child is pruned only against the nearer child's own result, and
// BBox::hit is a line test (student/bbox.cpp).  On the Cornell box a ray enters 5.9 of the 8 objects
// and tests 11.1 of the 12 triangles.  A per-lane walk therefore spends its time in divergent copies
// of the same heavy leaf code (ray -> object space, triangle / sphere tests, hit -> world space).
// Here that work is done WAVE-UNIFORMLY instead:
//
//   top-down sweep   over the interior nodes of the top-level tree in index order (parents first, a uniform
//            loop; the two child boxes arrive through the scalar unit): every lane runs BBox::hit for both
//            children of node q with the `times` its parent handed down, decides nearer / farther child and
//            hands (cur_close_t | cur_far_t) down to interior children through a per-lane LDS slot.
//   bottom-up sweep  over the same nodes in reverse: a leaf child is evaluated on the spot — all 64 lanes
//            transform their up-to-3 rays into the object's space (the three rays of a bounce share their
//            origin, so it is transformed once; matrices and triangles are scalar loads), run Sphere::hit or
//            the ordered Triangle::hit fold and compute the world distance Trace::transform recomputes —
//            an interior child's result is read back from its LDS slot; then the reference's visit rule
//            `cur_far_t.x < ret.distance || (!ret.hit && hitboth)` and Trace::min combine the two.
//   find_closest_hit is a pure function of (node, times), so evaluating every node — including subtrees
//   the recursion would have skipped, whose results are then simply not selected — gives the identical
//   answer with no per-lane control flow at all.  Only meshes with a real BVH<Triangle> (more than one
//   leaf) fall back to a per-lane walk inside the uniform object loop.
//
// Lanes run cycles in lockstep; a cycle traces one batch of three rays that share their origin:
//   bounce batch  {BSDF-sampled direct ray, MIS direct ray, indirect ray} of the lane's current path, or
//   camera burst  the camera rays of the lane's next three samples (same pixel, consecutive sample indices).
// After a burst the first sample's path goes on immediately, the other two camera hits are parked (one
// packed dword each) and picked up — without another trace — as soon as the current path ends.  All three
// slots of (almost) every batch therefore carry useful rays.  Work units are groups of three samples pulled
// from a global queue (one atomic per 64 units per wave; the last few samples of every pixel are queued as
// single-sample units at the very end to keep the tail of a launch short): persistent threads with per-lane path
// regeneration, no tail of idle lanes.  Every sample's radiance is written to a
// per-unit buffer; pt_reduce_kernel then adds the samples of a pixel in sample order, exactly as do_trace
// does (rays/pathtracer.cpp:216-226), so the image does not depend on which lane traced what.
#ifndef SRT_PT_WAVE_H
#define SRT_PT_WAVE_H


#include "pt_trace.h"
#include "pt_flat.h"
#include "pt_stream.h"

namespace srt {

constexpr uint32_t kWaveMaxObjects = 16;
constexpr uint32_t kChunk = 64;           // units (sample triples) a wave reserves per queue atomic (WaveParams::chunk)
constexpr uint32_t kBurst = 3;            // camera rays per burst = samples per unit (2 in the two-ray build)
constexpr uint32_t kMissTri = 0xFFFFFFFFu;
constexpr uint32_t kFlatReady = 16;        // TRAV 2: lanes with a finished batch that make the wave leave the walk
constexpr int kRecFields = 8;             // two float4 per bounce: {direct rgb, discrete} and {atten rgb, inv_pdf}

struct WaveParams {
  TileMap T;
  uint64_t seed;
  uint32_t sample_base;      // first sample index of this launch
  uint32_t samples;          // samples per pixel in this launch
  uint32_t groups3, singles; // units per pixel: groups3 full bursts of kBurst samples, then `singles` one-sample units
  uint32_t units3;           // pixels * groups3: units below this index are bursts
  uint32_t total_units;      // pixels * (groups3 + singles)
  uint32_t nlanes;           // threads of the whole grid (record scratch stride)
  float* sample_out;         // [unit] float4 {r, g, b, 0}: one aligned 16-byte store per finished sample
  float* records;            // float4 [(lane * kMaxPathDepth + level) * 2 + half]: a lane's records are one contiguous run (a bounce = 32 bytes,
                             // four bounces = one 128-byte line), so the read-back at the end of a path fetches the lines it uses and nothing else
                             // (half 0 = {direct rgb, discrete} when the bounce's direct light is known, half 1 = {atten rgb, 1/pdf} when it is shaded)
  unsigned long long* queue_head;   // next unit to hand out (zeroed before every launch)
  unsigned long long* ray_counter;  // scene.hit calls as the reference issues them, accumulated across launches
  unsigned long long* elided_counter;  // of those, the rays the two-ray build did not have to trace
  unsigned long long* stamps;
  uint32_t chunk;            // units a wave reserves per queue atomic
  uint32_t npix;             // pixel slots of the launch; sample_out is [sample][pixel slot] so that the reduction reads coalesced
  uint32_t flat_ready;       // TRAV 2: lanes with a finished batch that make the wave leave the walk
  uint32_t flat_interior;    // TRAV 2: lanes at interior nodes that keep the wave in the interior-step loop       // STAMP build only: per-section cycle sums
  // TRAV 3 (streamed form, pt_stream.h): one invocation = one generation; nlanes = path slots
  uint32_t* state;           // [word][slot] saved path state
  float4* ray_o;             // rays / walk requests of this generation at FIXED positions: [2 * ([queue slot][path slot])], and
  float4* ray_d;             //   ray_d = ray_o + 1 (origin and direction of a request side by side: one 32-byte piece);
                             //   TRAV 3: queue slot = batch slot; TRAV 4: queue slot = mesh ordinal * NR + batch slot
  const uint2* hits;         // [queue slot][path slot] results of the previous generation's rays / walks
  StreamCounters* sc;
  unsigned long long* block_counters;   // [block][2]: rays counted / rays elided by that block of the logic kernel, all generations
  uint32_t gen;              // generation number
  uint32_t obj_shift;        // packed hit = object slot << obj_shift | triangle
  // srt_pt_cancel (Pathtracer::cancel, rays/pathtracer.cpp:282-290 - the reference tests cancel_flag after every sample, :224):
  // host_cancel is the context's flag in pinned, device-visible host memory; dev_cancel the stream's sticky copy in device
  // memory, which every wave of every kernel of the epoch reads once, at its start.  A persistent launch gives up ONE of its
  // workgroups (0.1 % of its lanes) as the WATCHER: its first lane looks at the host word every few microseconds and, on seeing
  // it set, raises dev_cancel and moves the unit queue's head past its end - every tracing wave then finds the queue drained at
  // its next fetch.  Nothing in the tracing waves' loop knows about cancels: a poll in their fetch path, executed by one wave
  // only, still cost the loop 0.8 % (A/B in round 4: the branch alone changed the register allocation), and a DMA transfer from
  // the host onto the queue head never reached the launch (its atomics run on the line the L2 holds).
  const uint32_t* host_cancel;
  uint32_t* dev_cancel;
  uint32_t watch;            // persistent launches: the last workgroup is the cancel watcher (needs >= 2 workgroups)
  // streamed forms: the alive slots of the previous generation in ascending order (pt_compact_kernel); lane i of generation g >= 1
  // works on slot alive_list[i], i < sc->alive_n[g & 1].  Generation 0 (every slot idle) works on slot = lane.
  const uint32_t* alive_list;
};
constexpr unsigned long long kQueuePoison = 1ull << 62;
SRT_DEV bool cancel_raised(const uint32_t* dev_cancel) { return __hip_atomic_load(dev_cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; }
SRT_DEV bool cancel_requested(const uint32_t* host_cancel) { return __hip_atomic_load(host_cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u; }

constexpr uint32_t kStreamBlock = 256;    // threads per block of the streamed logic kernels (one queue atomic per block; four waves, so
                                          // that a CU takes the next block as soon as four waves are done)
constexpr uint32_t kMaxStreamSlots = 1u << 26;   // srt_pt_set_stream_slots: upper bound of the path-slot population (state planes: ~13 GB there)
constexpr uint32_t kMaxLazy = 4;          // TRAV 4: meshes with a real BVH<Triangle> whose walks are queued

// Wave-uniform launch constants passed through an empty asm: the value stays in SGPRs, but arithmetic on it
// (integer-division reciprocals, int->float conversions, matrix * constant products) can no longer be hoisted out
// of the persistent loop into VGPRs, where it would sit for the whole kernel or be spilled to scratch.
SRT_DEV uint32_t opq(uint32_t v) { asm volatile("" : "+s"(v)); return v; }
SRT_DEV float opq(float v) { asm volatile("" : "+s"(v)); return v; }
SRT_DEV uint64_t opq(uint64_t v) { asm volatile("" : "+s"(v)); return v; }
SRT_DEV TileMap opq(const TileMap& t) {
  return TileMap{opq(t.tile_w), opq(t.tile_h), opq(t.tiles_x), opq(t.tiles_y), opq(t.rank), opq(t.world), opq(t.local_tiles)};
}
SRT_DEV Camera opq(const Camera& c) {
  Camera o;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) o.iview.c[i][j] = opq(c.iview.c[i][j]);
  o.vert_fov = c.vert_fov; o.aspect_ratio = c.aspect_ratio;
  o.screen_h = opq(c.screen_h); o.screen_w = opq(c.screen_w);
  return o;
}

// pixel of local pixel index p (tile-major, 8x8 blocks inside a tile; same order as pt_epoch_kernel)
SRT_DEV void unit_pixel(const TileMap& T, uint32_t p, uint32_t& x, uint32_t& y) {
  const uint32_t px_per_tile = T.tile_w * T.tile_h;
  const uint32_t local_tile = p / px_per_tile, in_tile = p % px_per_tile;
  const uint32_t blocks_x = T.tile_w / 8;
  const uint32_t blk = in_tile / 64, l = in_tile % 64;
  const uint32_t lx = (blk % blocks_x) * 8 + (l % 8), ly = (blk / blocks_x) * 8 + (l / 8);
  const uint32_t tile = T.rank + local_tile * T.world;
  x = (tile % T.tiles_x) * T.tile_w + lx;
  y = (tile / T.tiles_x) * T.tile_h + ly;
}
// offset (in pixels) of (x, y)'s slot inside the tile-major output of this rank; same layout as pt_epoch_kernel
SRT_DEV uint32_t tile_slot(const TileMap& T, uint32_t p) {
  const uint32_t px_per_tile = T.tile_w * T.tile_h;
  const uint32_t local_tile = p / px_per_tile, in_tile = p % px_per_tile;
  const uint32_t blocks_x = T.tile_w / 8;
  const uint32_t blk = in_tile / 64, l = in_tile % 64;
  const uint32_t lx = (blk % blocks_x) * 8 + (l % 8), ly = (blk / blocks_x) * 8 + (l / 8);
  return local_tile * px_per_tile + ly * T.tile_w + lx;
}

// Mat4 * Vec3 for wave-uniform matrices: x / 1.0f == x bit for bit, so the perspective divide is skipped
// when w == 1 in every lane (always the case for affine matrices and finite points).
SRT_DEV V3 mat_point_uniform(const Mat4& m, V3 v) {
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; j++) o[j] = ((m.c[0][j] * v.x + m.c[1][j] * v.y) + m.c[2][j] * v.z) + m.c[3][j] * 1.0f;
  if (__ballot(o[3] != 1.0f) == 0ull) return v3(o[0], o[1], o[2]);
  return v3(o[0] / o[3], o[1] / o[3], o[2] / o[3]);
}

// BBox::hit with the reciprocal direction hoisted (the reference recomputes the same 1/dir per box), in
// straight-line form: the two early-outs become one verdict, `times` is narrowed only on a hit.
SRT_DEV bool box_hit_inv(const float* __restrict__ bx, V3 o, V3 inv, float& tx, float& ty) {
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

constexpr uint32_t kRetMiss = 0xFFFFFFFFu;
constexpr uint32_t kRetMissEnv = 0xFFFFFFFEu;   // miss, and the environment light is visible along the (camera) ray
// (the sweep builds take <= 16 objects: 5 + 27 bits; the streamed build passes the scene's own split)
SRT_DEV uint32_t pack_ret(const Hit& h, uint32_t shift = 27u) { return h.hit ? ((h.obj << shift) | h.tri) : kRetMiss; }
SRT_DEV Hit unpack_ret(float dist, uint32_t id, uint32_t shift = 27u) {
  Hit h;
  h.hit = id != kRetMiss;
  h.dist = h.hit ? dist : 0.0f;
  h.obj = h.hit ? (id >> shift) : 0;
  h.tri = h.hit ? (id & ((1u << shift) - 1u)) : 0;
  return h;
}
SRT_DEV Hit no_hit() { Hit h; h.hit = false; h.dist = 0.0f; h.obj = 0; h.tri = 0; return h; }

// Object::hit of object slot k for the three rays of a batch (shared origin), wave-uniformly: hit flag, the
// world distance Trace::transform recomputes, and the winning triangle (global index).
// HAS_BLAS = false compiles the per-lane BVH<Triangle> walk out (the host picks that build when every mesh is a
// single leaf, e.g. the Cornell box): the walk's registers would otherwise halve the occupancy of the common path.
// QUEUE (TRAV 4, the streamed sweeps): the walk of a real BVH<Triangle> is not done here.  In the probe pass (emit) the rays
// that need it are written - in the mesh's object space - to their fixed queue positions and flagged in emit_mask; the
// ray-cast kernel walks them between two generations; the complete pass reads {world distance, triangle} back.
template <bool HAS_BLAS, int NR, bool QUEUE = false>
SRT_DEV void object_testN(const DScene& S, uint32_t k, V3 org, const V3* d, const float* rb0, const float* rb1,
                          Counters& cnt, bool* hit, float* dist, uint32_t* tri, const bool* need, uint32_t* cidx,
                          const WaveParams* QP = nullptr, uint32_t lane_global = 0, bool emit = false, uint32_t* emit_mask = nullptr) {
  const Object& o = S.objects[k];
  const bool xf = o.has_trans != 0;
  V3 oorg = org;
  V3 od[NR];
  float ob0[NR], ob1[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) { od[r] = d[r]; ob0[r] = rb0[r]; ob1[r] = rb1[r]; }
  if (xf) {                                           // Ray::transform with the shared origin
    oorg = mat_point_uniform(o.itrans, org);
    float n2[NR], dn[NR], num[NR][3], q[NR][3];
    bool nz[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const V3 rd = mat_rotate(o.itrans, d[r]);
      num[r][0] = rd.x; num[r][1] = rd.y; num[r][2] = rd.z;
      n2[r] = norm2(rd);
      nz[r] = false;
    }
    sqrtN<NR>(n2, nz, dn);
    divNx3<NR, false>(num, dn, q);
#pragma unroll
    for (int r = 0; r < NR; r++) {
      ob0[r] *= dn[r]; ob1[r] *= dn[r];
      od[r] = v3(q[r][0], q[r][1], q[r][2]);
    }
  }
  V3 pos[NR];
  if (o.kind == OBJ_SPHERE) {
#pragma unroll
    for (int r = 0; r < NR; r++) {
      Ray ray;
      ray.o = oorg; ray.d = od[r]; ray.b0 = ob0[r]; ray.b1 = ob1[r];
      const SphHit sh = sphere_hit(o.radius, ray);
      hit[r] = sh.hit; tri[r] = 0;
      pos[r] = ray_at(ray, sh.t);
    }
    float n2[NR], nr[NR];
    bool miss[NR];                                    // no hit: t = 0, pos == origin
#pragma unroll
    for (int r = 0; r < NR; r++) { n2[r] = norm2(pos[r] - oorg); miss[r] = !hit[r]; }
    sqrtN<NR>(n2, miss, nr);
#pragma unroll
    for (int r = 0; r < NR; r++) dist[r] = fabsf(nr[r]);
  } else if (HAS_BLAS && QUEUE && o.use_bvh && o.nrec > 0) {
    const uint32_t m = o.use_bvh >> 8;                   // ordinal of this mesh (pt_scene.h)
#pragma unroll
    for (int r = 0; r < NR; r++) {
      hit[r] = false; dist[r] = 0.0f; tri[r] = 0u;
      if (need[r]) {
        const size_t pos = (size_t)(m * (uint32_t)NR + (uint32_t)r) * QP->nlanes + lane_global;
        if (emit) {
          QP->ray_o[2 * pos] = make_float4(oorg.x, oorg.y, oorg.z, ob0[r]);
          QP->ray_d[2 * pos] = make_float4(od[r].x, od[r].y, od[r].z, ob1[r]);
          *emit_mask |= 1u << (m * (uint32_t)NR + (uint32_t)r);
        } else {
          const uint2 hv = QP->hits[pos];
          hit[r] = hv.y != 0xFFFFFFFFu;
          dist[r] = hit[r] ? __uint_as_float(hv.x) : 0.0f;
          tri[r] = hit[r] ? hv.y : 0u;
        }
      }
    }
    return;                                              // world distances are final
  } else if (HAS_BLAS && !QUEUE && o.use_bvh && o.nrec > 0) {
    // A real BVH<Triangle>: a per-lane walk, the expensive leaf.  Only the rays whose traversal can reach this object
    // (need[], from the caller) are walked, and they are compacted over the wave first: the (lane, slot) pairs get
    // consecutive item numbers (ballot prefix), each lane then walks item `round * 64 + lane` - fetched from its owner
    // with ds_bpermute - and the owners collect the results the same way.  A batch of NR x 64 slots of which a third
    // needs the mesh costs one walk instead of NR.
    const int lane = (int)(threadIdx.x & 63u);
    uint32_t pos_[NR], total = 0;
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const unsigned long long m = __ballot(need[r]);
      pos_[r] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      total += (uint32_t)__popcll(m);
      hit[r] = false; dist[r] = 0.0f; tri[r] = 0u;
    }
#pragma unroll
    for (int r = 0; r < NR; r++)
      if (need[r]) cidx[pos_[r]] = (uint32_t)lane | ((uint32_t)r << 6);
    for (uint32_t base = 0; base < total; base += 64u) {
      const uint32_t item = base + (uint32_t)lane;
      const bool valid = item < total;
      const uint32_t code = valid ? cidx[item] : (uint32_t)lane;
      const int src = (int)(code & 63u), rr = (int)(code >> 6);
      Ray ray;
      ray.o = v3(__shfl(oorg.x, src), __shfl(oorg.y, src), __shfl(oorg.z, src));
      ray.d = v3(__shfl(od[0].x, src), __shfl(od[0].y, src), __shfl(od[0].z, src));
      ray.b0 = __shfl(ob0[0], src); ray.b1 = __shfl(ob1[0], src);
#pragma unroll
      for (int r = 1; r < NR; r++) {                     // every lane reads the NR candidates of lane `src`, keeps its slot's
        const float x = __shfl(od[r].x, src), y = __shfl(od[r].y, src), z = __shfl(od[r].z, src);
        const float c0 = __shfl(ob0[r], src), c1 = __shfl(ob1[r], src);
        if (rr == r) { ray.d = v3(x, y, z); ray.b0 = c0; ray.b1 = c1; }
      }
      Hit mh = no_hit();
      float wd = 0.0f;
      if (valid) {
        mh = mesh_hit<false, true>(S, o, ray, cnt);      // the lanes of a round walk together (while-while form)
        wd = mh.dist;
        if (mh.hit && xf) {                              // Trace::transform: distance = |T*position - T*origin|
          const TriHit th = tri_hit(S.tris[mh.tri], ray);
          wd = norm(mat_point(o.trans, ray_at(ray, th.t)) - mat_point(o.trans, ray.o));
        }
      }
#pragma unroll
      for (int r = 0; r < NR; r++) {                     // results back to the owners
        const int from = (int)((pos_[r] - base) & 63u);
        const int h_ = __shfl(mh.hit ? 1 : 0, from);
        const float d_ = __shfl(wd, from);
        const uint32_t t_ = (uint32_t)__shfl((int)mh.tri, from);
        if (need[r] && pos_[r] >= base && pos_[r] < base + 64u) { hit[r] = h_ != 0; dist[r] = d_; tri[r] = t_; }
      }
    }
    return;                                              // world distances are final
  } else {                                            // one leaf of <= 4 triangles, or List<Triangle>: ordered fold
    bool bh[NR];
    float bd[NR], bt[NR];
    uint32_t bi[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) { bh[r] = false; bd[r] = 0.0f; bt[r] = 0.0f; bi[r] = 0u; }
    for (uint32_t t = 0; t < o.ntri; t++) {
      TriHit th[NR];
      tri_hitN<NR>(S.tris[o.tri_base + t], oorg, od, ob0, ob1, th);
#pragma unroll
      for (int r = 0; r < NR; r++) {
        const bool keep = left_wins(bh[r], bd[r], th[r].hit, th[r].dist);   // ret = Trace::min(ret, hit)
        bd[r] = keep ? bd[r] : (th[r].hit ? th[r].dist : 0.0f);
        bt[r] = keep ? bt[r] : (th[r].hit ? th[r].t : 0.0f);
        bi[r] = keep ? bi[r] : (th[r].hit ? o.tri_base + t : 0u);
        bh[r] = keep ? bh[r] : th[r].hit;
      }
    }
#pragma unroll
    for (int r = 0; r < NR; r++) {
      hit[r] = bh[r]; dist[r] = bd[r]; tri[r] = bi[r];
      pos[r] = v3(oorg.x + od[r].x * bt[r], oorg.y + od[r].y * bt[r], oorg.z + od[r].z * bt[r]);
    }
  }
  bool any = false;
#pragma unroll
  for (int r = 0; r < NR; r++) any = any || hit[r];
  if (xf && __ballot(any) != 0ull) {
    const V3 ow = mat_point_uniform(o.trans, oorg);   // Trace::transform: distance = |T*position - T*origin|
    float n2[NR], nr[NR];
    bool miss[NR];                                    // the same point twice: exactly +0
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const V3 p = hit[r] ? pos[r] : oorg;            // lanes without a hit transform a harmless point
      n2[r] = norm2(mat_point_uniform(o.trans, p) - ow);
      miss[r] = !hit[r];
    }
    sqrtN<NR>(n2, miss, nr);
#pragma unroll
    for (int r = 0; r < NR; r++) dist[r] = hit[r] ? nr[r] : dist[r];
  }
}


// TRAV 4 (the streamed sweeps): Object::hit of a mesh with a real BVH<Triangle> for the rays of a batch that need it.  Only the
// ray -> object space step (Ray::transform, as object_testN does it) happens here; the walk itself is queued: the probe pass
// (emit) writes the object-space rays to their fixed queue positions and flags them in emit_mask, the ray-cast kernel walks
// them between two generations, the complete pass reads {hit, world distance, triangle} back.
template <int NR>
SRT_DEV void object_queueN(const DScene& S, uint32_t k, V3 org, const V3* d, const float* rb0, const float* rb1, bool* hit, float* dist,
                           uint32_t* tri, const bool* need, const WaveParams& QP, uint32_t lane_global, bool emit, uint32_t& emit_mask) {
  const Object& o = S.objects[k];
  const uint32_t m = o.use_bvh >> 8;                     // ordinal of this mesh (pt_scene.h)
#pragma unroll
  for (int r = 0; r < NR; r++) { hit[r] = false; dist[r] = 0.0f; tri[r] = 0u; }
  if (!emit) {
#pragma unroll
    for (int r = 0; r < NR; r++) {
      if (need[r]) {
        const uint2 hv = nt_load_hit(QP.hits, (size_t)(m * (uint32_t)NR + (uint32_t)r) * QP.nlanes + lane_global);
        hit[r] = hv.y != 0xFFFFFFFFu;
        dist[r] = hit[r] ? __uint_as_float(hv.x) : 0.0f;
        tri[r] = hit[r] ? hv.y : 0u;
      }
    }
    return;
  }
  bool any = false;
#pragma unroll
  for (int r = 0; r < NR; r++) any = any || need[r];
  if (__ballot(any) == 0ull) return;
  V3 oorg = org;
  V3 od[NR];
  float ob0[NR], ob1[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) { od[r] = d[r]; ob0[r] = rb0[r]; ob1[r] = rb1[r]; }
  if (o.has_trans != 0) {                                 // Ray::transform with the shared origin (as in object_testN)
    oorg = mat_point_uniform(o.itrans, org);
    float n2[NR], dn[NR], num[NR][3], q[NR][3];
    bool nz[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const V3 rd = mat_rotate(o.itrans, d[r]);
      num[r][0] = rd.x; num[r][1] = rd.y; num[r][2] = rd.z;
      n2[r] = norm2(rd);
      nz[r] = false;
    }
    sqrtN<NR>(n2, nz, dn);
    divNx3<NR, false>(num, dn, q);
#pragma unroll
    for (int r = 0; r < NR; r++) {
      ob0[r] *= dn[r]; ob1[r] *= dn[r];
      od[r] = v3(q[r][0], q[r][1], q[r][2]);
    }
  }
  // The first step of the walk is taken here: BVH<Triangle>::find_closest_hit tests the root's two child boxes with the object-space
  // ray and returns "no hit" when both miss (student/bvh.inl:182-186).  The ray-cast kernel would find exactly that - same record,
  // same BBox::hit arithmetic (box_hit_rec), same reciprocal direction - after fetching the request, setting the walk up and
  // retiring it; a ray that misses both boxes gets its answer written straight into the hit plane and is never queued.  (A ray
  // is asked about the mesh whenever it hits either box of the TLAS node ABOVE the mesh's leaf; most of those pass the mesh by.)
  const WaveInterior& root = S.blas_recs[o.rec_base];
#pragma unroll
  for (int r = 0; r < NR; r++) {
    if (need[r]) {
      const size_t pos = (size_t)(m * (uint32_t)NR + (uint32_t)r) * QP.nlanes + lane_global;
      const V3 cinv = v3(1.0f / od[r].x, 1.0f / od[r].y, 1.0f / od[r].z);
      float ta = 0.0f, tb = 0.0f, tc = 0.0f, td = 0.0f;
      const bool hl = box_hit_rec(root.boxl, oorg, cinv, ta, tb);
      const bool hr = box_hit_rec(root.boxr, oorg, cinv, tc, td);
      if (hl || hr) {
        nt_store_ray(QP.ray_o, 2 * pos, oorg.x, oorg.y, oorg.z, ob0[r]);
        nt_store_ray(QP.ray_d, 2 * pos, od[r].x, od[r].y, od[r].z, ob1[r]);
        emit_mask |= 1u << (m * (uint32_t)NR + (uint32_t)r);
      } else {
        __builtin_nontemporal_store(u32x2_t{0u, 0xFFFFFFFFu}, reinterpret_cast<u32x2_t*>(const_cast<uint2*>(QP.hits)) + pos);
      }
    }
  }
}

// Scene arrays are passed as separate `const T* __restrict__` kernel arguments (not inside DScene): only then can
// the compiler prove that the stores to records / sample_out do not clobber them and turn the wave-uniform
// scene reads into scalar loads (s_load_*), which is what keeps the sweeps off the vector memory path.
//
// STAMP = true is a diagnostic build: s_memtime deltas of the loop's sections are summed per wave and added to
// P.stamps (never used for results or for reported times; the stamps themselves perturb the schedule).
enum { ST_REFILL = 0, ST_TOPDOWN, ST_LEAVES, ST_COMBINE, ST_POST, ST_SHADE, ST_TERMINATE, ST_COUNT_ };
// NR = rays per batch.  3: the batch described above.  2: the build for scenes without delta / environment lights whose
// continuous BSDFs are all Lambertian.  There the BSDF-sampled direct ray of sample_direct_lighting is dead: the reference
// adds its term and subtracts it again (student/pathtracer.cpp:118-125: radiance = (point_lighting + direct) - direct with
// point_lighting == 0, i.e. +0 for any finite direct, and a NaN direct comes with a NaN second term), so only its random
// draws are kept and a bounce batch is {MIS direct ray | the direct ray of a discrete BSDF, indirect ray}; units are
// pairs of samples.  Rays are counted as the reference issues them (P.ray_counter); P.elided_counter counts the ones not traced.
// A dword of per-lane path state that is touched a few times per path (pixel, sample window, parked camera hits).  The
// persistent sweep kernel keeps these in LDS, [variable][thread]: it needs ~150 VGPRs and runs with 128 (4 waves / SIMD is
// worth 11 % over 3), so what does not fit goes to scratch, and scratch lines that fall out of L2 were most of the kernel's
// memory traffic (profiles/README.md); a value parked in LDS costs a ds_read where it is used and no traffic at all.
template <bool IN_LDS> struct ColdU32;
template <> struct ColdU32<false> {
  uint32_t v;
  SRT_DEV void bind(uint32_t*) {}
  SRT_DEV operator uint32_t() const { return v; }
  SRT_DEV ColdU32& operator=(uint32_t x) { v = x; return *this; }
  SRT_DEV ColdU32& operator=(const ColdU32& o) { v = o.v; return *this; }
};
template <> struct ColdU32<true> {
  uint32_t* p;
  SRT_DEV void bind(uint32_t* at) { p = at; }
  SRT_DEV operator uint32_t() const { return *p; }
  SRT_DEV ColdU32& operator=(uint32_t x) { *p = x; return *this; }
  SRT_DEV ColdU32& operator=(const ColdU32& o) { *p = *o.p; return *this; }   // (the VALUE: the default would rebind)
};
// Four words per lane is what the LDS has left at 16 waves / CU next to the sweep slots of the Cornell box's seven-node BVH<Object>:
// pixel (x | y << 16), local pixel index, sample window (first | current << 16 | count << 30), the first parked camera hit.
constexpr int kColdWords = 4;
__host__ __device__ constexpr bool cold_in_lds(int trav) { return trav == 0; }

// PHASE (streamed sweeps, TRAV 4, without delta / environment lights): the two passes of a generation as TWO kernels.
//   1 "resolve": load the slot, complete sweep with the walks' results, finish the bounce, terminate or shade, save.
//   2 "probe":   load what the next batch's rays need (origin, directions, bounds, flags), refill idle slots from the unit queue,
//                box-test the nodes above the queued meshes, emit the walk requests, save the emit mask (and a new unit's words).
//   0: both passes in one kernel (TRAV 3, and the DL build of TRAV 4).
// One kernel carrying a slot's whole state through both sweeps wanted ~245 VGPRs at four waves per SIMD (121 spilled, ~150 GB of
// scratch traffic per 64-spp epoch of BASELINE configs[4]); the resolve kernel alone is the Cornell kernel's loop body, and the
// probe kernel needs neither the sweep slots in LDS nor most of the state.
template <bool STAMP, int TRAV, bool DL, int NR, int PHASE = 0>
#ifndef SRT_WAVE_OCC
#define SRT_WAVE_OCC 4
#endif
#ifndef SRT_WAVE_OCC2
#define SRT_WAVE_OCC2 5
#endif
#ifndef SRT_WAVE_OCC2T
#define SRT_WAVE_OCC2T 4
#endif
#ifndef SRT_WAVE_OCC3T
#define SRT_WAVE_OCC3T 4
#endif
#ifndef SRT_STREAM_OCC
#define SRT_STREAM_OCC 4
#endif
#ifndef SRT_PROBE_OCC
#define SRT_PROBE_OCC 8
#endif
__global__ __launch_bounds__(256, PHASE == 2 ? SRT_PROBE_OCC : TRAV == 4 ? SRT_STREAM_OCC : (NR == 2 ? (TRAV == 1 ? SRT_WAVE_OCC2T : (TRAV >= 3 ? 4 : SRT_WAVE_OCC2)) : (TRAV == 1 ? SRT_WAVE_OCC3T : SRT_WAVE_OCC))) void pt_wave_kernel(DScene S_in, WaveParams P_in, const Object* __restrict__ a_objects,
                                                      const Tri* __restrict__ a_tris, const TriNrm* __restrict__ a_nrm,
                                                      const Node* __restrict__ a_nodes, const Light* __restrict__ a_lights,
                                                      const LightTri* __restrict__ a_ltris, const Material* __restrict__ a_mats,
                                                      const WaveInterior* __restrict__ a_wave, const WaveInterior* __restrict__ a_blas,
                                                      float* __restrict__ a_records, float* __restrict__ a_samples) {
  static_assert(NR == 3 || (NR == 2 && TRAV != 2 && !DL), "two-ray batches: sweep builds without delta / environment lights only");
  static_assert(!(STAMP && TRAV >= 3), "the streamed builds have no section stamps");
  static_assert(PHASE == 0 || (TRAV == 4 && !DL), "split generations: the streamed sweeps without delta / environment lights");
  constexpr int C = NR - 1;                              // slot of the indirect ray
  constexpr bool STREAM = TRAV >= 3;                     // pt_stream.h: 3 = every ray through the ray-cast kernel, 4 = sweeps here, BVH<Triangle> walks queued
  constexpr bool LAZY = TRAV == 1 || TRAV == 4;          // meshes with a real BVH<Triangle> are evaluated lazily inside the sweeps
  if constexpr (STREAM) {
    if (P_in.sc->done != 0u) return;                     // every unit is finished: the remaining generations are no-ops
  }
  if (cancel_raised(P_in.dev_cancel)) return;            // srt_pt_cancel: what is left of the epoch is not rendered
  if constexpr (!STREAM) {
    // The cancel watcher (WaveParams::watch): the launch's last workgroup traces nothing.  Its first lane sleeps, looks at the host's
    // flag, sleeps - until the flag is up (then: the stream's cancel word, and the queue head past its end) or every unit has been
    // handed out (a cancel later than that has nothing left to stop).  Both ends are reached whatever the other workgroups do.
    if (P_in.watch != 0u && blockIdx.x == gridDim.x - 1u) {
      if (threadIdx.x != 0u) return;
      for (;;) {
        if (cancel_requested(P_in.host_cancel)) {
          atomicExch(P_in.dev_cancel, 1u);
          atomicMax(P_in.queue_head, kQueuePoison);
          return;
        }
        if (__hip_atomic_load(P_in.queue_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)P_in.total_units) return;
        __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(127);
      }
    }
  }
  DScene S = S_in;
  S.objects = a_objects; S.tris = a_tris; S.tri_nrm = a_nrm; S.nodes = a_nodes; S.lights = a_lights;
  S.light_tris = a_ltris; S.materials = a_mats; S.wave_tlas = a_wave; S.blas_recs = a_blas;
  WaveParams P = P_in;
  P.records = a_records; P.sample_out = a_samples;
  extern __shared__ float lds_f[];
  // item table of the compacted per-lane walks (object_testN): one dword per batch slot of the wave
  __shared__ uint32_t s_cidx[TRAV == 1 ? 4 * NR * 64 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* const cidx = s_cidx + (TRAV == 1 ? wave * NR * 64 : 0);
  const uint32_t nobj = S.nobjects;
  const uint32_t Q = (S.use_bvh && TRAV != 3) ? S.wave_q : 0u;
  const uint32_t oshift = (TRAV == 3) ? P.obj_shift : 27u;   // packed hit = object slot << oshift | triangle
  __shared__ unsigned long long s_blk[STREAM ? 36 : 1];  // streamed builds: per-wave partial sums of the block-wide exchanges
  unsigned long long stamp_acc[ST_COUNT_] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_t = 0;
  if (STAMP) stamp_t = __builtin_readcyclecounter();
#define SECTION_END(i)                                                        \
  if (STAMP) {                                                                \
    const unsigned long long now_ = __builtin_readcyclecounter();             \
    stamp_acc[i] += now_ - stamp_t;                                           \
    stamp_t = now_;                                                           \
  }
  // per-wave sweep slots: [q][ray][field] x 64 lanes.  A slot is reused as it goes through the sweeps:
  //   written by the parent   field 0 = tin.x, field 1 = tin.y      (top-down, before step q)
  //   after step q            field 0 = cur_far_t.x of node q        (tin is dead once read)
  //   after bottom-up step q  field 0 = ret.dist, field 1 = ret.id   (read by the parent)
  // The root (q = 0) has no slot: it receives no `times` and nobody reads its result; its cur_far_t.x stays in registers.
  constexpr bool COLD = cold_in_lds(TRAV);
  uint32_t* const cold = reinterpret_cast<uint32_t*>(lds_f) + threadIdx.x;           // [variable][thread of the block]
  float* wl = lds_f + (COLD ? kColdWords * 256 : 0) + (size_t)wave * (Q > 0 ? Q - 1 : 0) * (2 * NR) * 64 + lane;   // (PHASE 2: no dynamic LDS, never touched)
#define SLOT(q, r, f) wl[((((q) - 1) * NR + (r)) * 2 + (f)) * 64]

  const uint32_t lane_phys = blockIdx.x * blockDim.x + threadIdx.x;
  // `lane_global` indexes everything a path slot owns (state planes, records, ray / hit planes).  Persistent kernels: the lane itself.
  // Streamed kernels: the slot this lane serves in this generation - taken from the alive list once the launch has one.
  uint32_t lane_global = lane_phys;
  bool has_slot = true;
  if constexpr (STREAM) {
    if (P.gen != 0u) {
      const uint32_t n_alive = P.sc->alive_n[P.gen & 1u];
      if (PHASE != 2 && lane_phys == 0u) { P.sc->nrays[(P.gen + 1u) & 1u] = 0u; P.sc->cast_head[(P.gen + 1u) & 1u] = 0u; P.sc->alive[(P.gen + 1u) & 1u] = 0u; P.sc->alive_n[(P.gen + 1u) & 1u] = 0u; }
      if (blockIdx.x * blockDim.x >= n_alive) return;    // (a whole block without a live slot: nothing to do - uniform, before any barrier)
      has_slot = lane_phys < n_alive;
      lane_global = has_slot ? P.alive_list[lane_phys] : 0u;
    }
  }
  Counters cnt;
  cnt.v[C_RAYS] = 0;
  uint32_t traced = 0;                                   // rays this lane actually traced (== cnt.v[C_RAYS] when NR == 3)

  // ---- persistent per-lane path state ----
  bool alive = false;                                    // the lane owns a unit that is not finished yet
  bool burst = false;                                    // the batch in flight is a camera burst (else a bounce batch)
  ColdU32<COLD> pxy;                                     // pixel of the unit: x | y << 16 (srt_pt_set_params: both < 65536)
  ColdU32<COLD> samp;                                    // samples [first, first + count) of the launch and the current one:
                                                         // first | current << 16 | count << 30 (the SW_SAMPLES word)
  ColdU32<COLD> pixel_slot;                              // local pixel index (sample_out addressing)
  ColdU32<COLD> pend0;                                   // parked camera hits of samples current+1, current+2 (burst order):
  uint32_t pend1 = kRetMiss;                             // the first in LDS with the others, the second (NR == 3) in a register
  pxy.bind(cold + 0 * 256); samp.bind(cold + 1 * 256); pixel_slot.bind(cold + 2 * 256); pend0.bind(cold + 3 * 256);
  pxy = 0u; samp = 0u; pixel_slot = 0u; pend0 = kRetMiss;
#define PX_ ((uint32_t)pxy & 0xffffu)
#define PY_ ((uint32_t)pxy >> 16)
#define S_FIRST_ ((uint32_t)samp & 0xffffu)
#define S_CUR_ (((uint32_t)samp >> 16) & 0x3fffu)
#define S_COUNT_ ((uint32_t)samp >> 30)
  uint32_t depth = 0, level = 0;
  Rng rng;
  rng.state = 0; rng.inc = 1; rng.draws = 0;
  V3 org = v3(0, 0, 0);
  V3 d[NR];                                              // bounce: A BSDF direct, B MIS direct, C indirect (NR == 2: direct, indirect); burst: camera rays
#pragma unroll
  for (int j = 0; j < NR; j++) d[j] = v3(0.25f, 0.5f, 0.75f);
  float cb0 = 0.0f, cb1 = 0.0f;                          // dist_bounds shared by the batch: [EPS_F, FLT_MAX] or the camera's [0, inf]
  bool actA = false, actB = false;                       // bounce batch: slots in use (C always); burst: slots 1, 2 in use
  Spec att = spec(0, 0, 0);                              // s1.attenuation (== evaluate(out) for Lambertian)
  float pdf4 = 1.0f, pdf_area = 0.0f;
  bool discrete = false;

  // DL (scenes with delta lights, Pathtracer::point_lighting): after a continuous-BSDF bounce's A/B/C batch the lane
  // traces the shadow rays of the lights, three per batch, from the same origin; the bounce's direct term is
  // completed when the last shadow batch returns, then the held C hit decides how the path goes on
  bool sh_phase = false, sa1 = false, sa2 = false;       // a shadow batch is in flight; its slots 1, 2 carry a ray
  uint32_t light_i = 0, held_chit = kRetMiss;
  Spec pl = spec(0, 0, 0), dA_keep = spec(0, 0, 0), d6_keep = spec(0, 0, 0);
  V3 dC_keep = v3(0, 0, 1);                              // direction of the C ray (the next shade needs it)
  float sb1[3] = {0.0f, 0.0f, 0.0f};                     // dist_bounds.y of the shadow rays: distance - EPS_F

  // TRAV == 2: the flattened walk's per-lane state and stack persist across iterations of the loop below, so the
  // wave can stop walking once kFlatReady lanes have a finished batch, shade / refill those, and resume the rest
  FlatState FS;
  FlatFrame fstack[TRAV == 2 ? kFlatStack : 1];
  bool need_begin = false;                               // a new batch (camera burst or bounce) waits to be started

  // wave-uniform queue window
  uint32_t chunk_next = 0, chunk_end = 0;
  bool queue_empty = false;
  uint32_t units_finished = 0;                           // TRAV 3: units this lane completed (or found outside the image)

  // TRAV == 3 (pt_stream.h): this invocation is one generation.  The slot's state comes from HBM, pass 0 of the loop
  // consumes the hits of the batch emitted by the previous generation (step 3 as it stands), pass 1 refills idle slots,
  // appends the next batch's rays to the queue and saves the state.
#define ST(w) __builtin_nontemporal_load(P.state + ((size_t)(w) * P.nlanes + lane_global))
#define ST_SET(w, v) __builtin_nontemporal_store((uint32_t)(v), P.state + ((size_t)(w) * P.nlanes + lane_global))
// (non-temporal: a record is written once and read once, a sample is written once - they should not push the waves' scratch
//  lines out of L2 on their way through)
#define REC_AT(lev, half) (reinterpret_cast<f32x4*>(P.records) + (((size_t)lane_global * kMaxPathDepth + (size_t)(lev)) * 2u + (half)))
#define REC_STORE(lev, half, x, y, z, w) __builtin_nontemporal_store(f32x4{x, y, z, w}, REC_AT(lev, half))
#define REC_LOAD(lev, half) (*REC_AT(lev, half))
  uint32_t emit_mask = 0;                                // queue slots of this lane that carry a ray / walk request for the next cast
  if constexpr (STREAM && PHASE == 2) {
    // the probe kernel: what the batch's rays need, nothing else
    const uint32_t fw = has_slot ? ST(SW_FLAGS) : 0u;
    alive = (fw & 1u) != 0;
    if (alive) {
      burst = (fw & 2u) != 0; actA = (fw & 4u) != 0; actB = (fw & 8u) != 0;
      org = v3(__uint_as_float(ST(SW_ORG)), __uint_as_float(ST(SW_ORG + 1)), __uint_as_float(ST(SW_ORG + 2)));
      d[C] = v3(__uint_as_float(ST(SW_DC)), __uint_as_float(ST(SW_DC + 1)), __uint_as_float(ST(SW_DC + 2)));
      cb0 = __uint_as_float(ST(SW_CB0)); cb1 = __uint_as_float(ST(SW_CB1));
      d[0] = v3(__uint_as_float(ST(SW_D0)), __uint_as_float(ST(SW_D0 + 1)), __uint_as_float(ST(SW_D0 + 2)));
      if (NR > 2) d[1] = v3(__uint_as_float(ST(SW_D1)), __uint_as_float(ST(SW_D1 + 1)), __uint_as_float(ST(SW_D1 + 2)));
    }
  } else if constexpr (STREAM) {
    if (P.gen == 0u && lane_phys == 0u) { P.sc->nrays[1] = 0u; P.sc->cast_head[1] = 0u; P.sc->alive[1] = 0u; P.sc->alive_n[1] = 0u; }   // the next generation's (later generations: above)
    const uint32_t fw = has_slot ? ST(SW_FLAGS) : 0u;
    alive = (fw & 1u) != 0;
    if (alive) {
      burst = (fw & 2u) != 0; actA = (fw & 4u) != 0; actB = (fw & 8u) != 0; discrete = (fw & 16u) != 0;
      sh_phase = (fw & 32u) != 0; sa1 = (fw & 64u) != 0; sa2 = (fw & 128u) != 0;
      level = (fw >> 8) & 0xffu; depth = (fw >> 16) & 0xffu;
      pxy = ST(SW_PX) | (ST(SW_PY) << 16); pixel_slot = ST(SW_PIXEL_SLOT);
      samp = ST(SW_SAMPLES);
      pend0 = ST(SW_PEND0);
      if (NR > 2) pend1 = ST(SW_PEND1);
      rng.state = (uint64_t)ST(SW_RNG_LO) | ((uint64_t)ST(SW_RNG_HI) << 32);
      rng.inc = (((((uint64_t)(PY_ * S.w + PX_)) << 32) | (uint64_t)(P.sample_base + S_CUR_)) << 1) | 1ull;   // Rng::key's increment
      org = v3(__uint_as_float(ST(SW_ORG)), __uint_as_float(ST(SW_ORG + 1)), __uint_as_float(ST(SW_ORG + 2)));
      d[C] = v3(__uint_as_float(ST(SW_DC)), __uint_as_float(ST(SW_DC + 1)), __uint_as_float(ST(SW_DC + 2)));
      cb0 = __uint_as_float(ST(SW_CB0)); cb1 = __uint_as_float(ST(SW_CB1));
      att = spec(__uint_as_float(ST(SW_ATT)), __uint_as_float(ST(SW_ATT + 1)), __uint_as_float(ST(SW_ATT + 2)));
      pdf4 = __uint_as_float(ST(SW_PDF4)); pdf_area = __uint_as_float(ST(SW_PDF_AREA));
      if constexpr (DL || TRAV == 4) {                   // (TRAV 4 traces the batch again in the complete pass: every direction)
        d[0] = v3(__uint_as_float(ST(SW_D0)), __uint_as_float(ST(SW_D0 + 1)), __uint_as_float(ST(SW_D0 + 2)));
        if (NR > 2) d[1] = v3(__uint_as_float(ST(SW_D1)), __uint_as_float(ST(SW_D1 + 1)), __uint_as_float(ST(SW_D1 + 2)));
      }
      if constexpr (DL) {
        light_i = ST(SW_LIGHT_I); held_chit = ST(SW_HELD);
        pl = spec(__uint_as_float(ST(SW_PL)), __uint_as_float(ST(SW_PL + 1)), __uint_as_float(ST(SW_PL + 2)));
        dA_keep = spec(__uint_as_float(ST(SW_DA)), __uint_as_float(ST(SW_DA + 1)), __uint_as_float(ST(SW_DA + 2)));
        d6_keep = spec(__uint_as_float(ST(SW_D6)), __uint_as_float(ST(SW_D6 + 1)), __uint_as_float(ST(SW_D6 + 2)));
        dC_keep = v3(__uint_as_float(ST(SW_DCK)), __uint_as_float(ST(SW_DCK + 1)), __uint_as_float(ST(SW_DCK + 2)));
        sb1[0] = __uint_as_float(ST(SW_SB1)); sb1[1] = __uint_as_float(ST(SW_SB1 + 1)); sb1[2] = __uint_as_float(ST(SW_SB1 + 2));
      }
    }
  }

  bool refilled = false;                                 // PHASE 2: the slot took a new unit in this generation
  bool retry = false;                                    // streamed: the unit the slot took was a padding pixel - it must come back for another
  auto save_state = [&]() {
    if (!has_slot) return;                               // (a lane past the end of the alive list serves no slot)
    if constexpr (PHASE == 2) {
      // the probe kernel changes a slot's state only by giving it a new unit: the words the refill sets (the parked hits, the RNG
      // state and the shading terms are written by the resolve kernel before anything reads them)
      ST_SET(SW_EMIT, emit_mask | ((alive || retry) ? 0x80000000u : 0u));   // (bit 31: the slot stays on the alive list)
      if (refilled) {
        ST_SET(SW_FLAGS, 1u | 2u | (actA ? 4u : 0u) | (actB ? 8u : 0u) | (level << 8) | (depth << 16));
        ST_SET(SW_PX, PX_); ST_SET(SW_PY, PY_); ST_SET(SW_PIXEL_SLOT, pixel_slot);
        ST_SET(SW_SAMPLES, samp);
        ST_SET(SW_ORG, __float_as_uint(org.x)); ST_SET(SW_ORG + 1, __float_as_uint(org.y)); ST_SET(SW_ORG + 2, __float_as_uint(org.z));
        ST_SET(SW_DC, __float_as_uint(d[C].x)); ST_SET(SW_DC + 1, __float_as_uint(d[C].y)); ST_SET(SW_DC + 2, __float_as_uint(d[C].z));
        ST_SET(SW_CB0, __float_as_uint(cb0)); ST_SET(SW_CB1, __float_as_uint(cb1));
        ST_SET(SW_D0, __float_as_uint(d[0].x)); ST_SET(SW_D0 + 1, __float_as_uint(d[0].y)); ST_SET(SW_D0 + 2, __float_as_uint(d[0].z));
        if (NR > 2) { ST_SET(SW_D1, __float_as_uint(d[1].x)); ST_SET(SW_D1 + 1, __float_as_uint(d[1].y)); ST_SET(SW_D1 + 2, __float_as_uint(d[1].z)); }
      }
      return;
    }
    ST_SET(SW_FLAGS, (alive ? 1u : 0u) | (burst ? 2u : 0u) | (actA ? 4u : 0u) | (actB ? 8u : 0u) | (discrete ? 16u : 0u) |
                   (sh_phase ? 32u : 0u) | (sa1 ? 64u : 0u) | (sa2 ? 128u : 0u) | (level << 8) | (depth << 16));
    if constexpr (PHASE == 0) ST_SET(SW_EMIT, emit_mask | ((alive || retry) ? 0x80000000u : 0u));   // (bit 31: the slot stays on the alive list - alive, or it drew a padding pixel and must draw again; a generation without requests need not be the last; PHASE 1: the probe kernel writes it)
    if (alive) {
      ST_SET(SW_PX, PX_); ST_SET(SW_PY, PY_); ST_SET(SW_PIXEL_SLOT, pixel_slot);
      ST_SET(SW_SAMPLES, samp);
      ST_SET(SW_PEND0, pend0);
      if (NR > 2) ST_SET(SW_PEND1, pend1);
      ST_SET(SW_RNG_LO, (uint32_t)rng.state); ST_SET(SW_RNG_HI, (uint32_t)(rng.state >> 32));
      ST_SET(SW_ORG, __float_as_uint(org.x)); ST_SET(SW_ORG + 1, __float_as_uint(org.y)); ST_SET(SW_ORG + 2, __float_as_uint(org.z));
      ST_SET(SW_DC, __float_as_uint(d[C].x)); ST_SET(SW_DC + 1, __float_as_uint(d[C].y)); ST_SET(SW_DC + 2, __float_as_uint(d[C].z));
      ST_SET(SW_CB0, __float_as_uint(cb0)); ST_SET(SW_CB1, __float_as_uint(cb1));
      ST_SET(SW_ATT, __float_as_uint(att.r)); ST_SET(SW_ATT + 1, __float_as_uint(att.g)); ST_SET(SW_ATT + 2, __float_as_uint(att.b));
      ST_SET(SW_PDF4, __float_as_uint(pdf4)); ST_SET(SW_PDF_AREA, __float_as_uint(pdf_area));
      if constexpr (DL || TRAV == 4) {
        ST_SET(SW_D0, __float_as_uint(d[0].x)); ST_SET(SW_D0 + 1, __float_as_uint(d[0].y)); ST_SET(SW_D0 + 2, __float_as_uint(d[0].z));
        if (NR > 2) { ST_SET(SW_D1, __float_as_uint(d[1].x)); ST_SET(SW_D1 + 1, __float_as_uint(d[1].y)); ST_SET(SW_D1 + 2, __float_as_uint(d[1].z)); }
      }
      if constexpr (DL) {
        ST_SET(SW_LIGHT_I, light_i); ST_SET(SW_HELD, held_chit);
        ST_SET(SW_PL, __float_as_uint(pl.r)); ST_SET(SW_PL + 1, __float_as_uint(pl.g)); ST_SET(SW_PL + 2, __float_as_uint(pl.b));
        ST_SET(SW_DA, __float_as_uint(dA_keep.r)); ST_SET(SW_DA + 1, __float_as_uint(dA_keep.g)); ST_SET(SW_DA + 2, __float_as_uint(dA_keep.b));
        ST_SET(SW_D6, __float_as_uint(d6_keep.r)); ST_SET(SW_D6 + 1, __float_as_uint(d6_keep.g)); ST_SET(SW_D6 + 2, __float_as_uint(d6_keep.b));
        ST_SET(SW_DCK, __float_as_uint(dC_keep.x)); ST_SET(SW_DCK + 1, __float_as_uint(dC_keep.y)); ST_SET(SW_DCK + 2, __float_as_uint(dC_keep.z));
        ST_SET(SW_SB1, __float_as_uint(sb1[0])); ST_SET(SW_SB1 + 1, __float_as_uint(sb1[1])); ST_SET(SW_SB1 + 2, __float_as_uint(sb1[2]));
      }
    }
  };

  for (int pass = (PHASE == 2 ? 1 : 0);; pass++) {
    // ---------------- 1. refill idle lanes ----------------
    const unsigned long long need = (!STREAM || (PHASE != 1 && pass == 1)) ? __ballot(!alive && has_slot) : 0ull;
    uint32_t stream_unit = kMissTri;
    if constexpr (STREAM && PHASE != 1) {
      if (pass == 1) {
        // one queue atomic per BLOCK: the waves' wants meet in LDS (a single word saturates at ~90 atomics / us, and a
        // generation of a million slots would bring sixteen thousand of them)
        const uint32_t want = (uint32_t)__popcll(need);
        const uint32_t my_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
        if (lane == 0) s_blk[wave] = want;
        __syncthreads();
        if (threadIdx.x == 0) {
          unsigned long long total = 0;
          for (uint32_t w = 0; w < blockDim.x / 64u; w++) { const unsigned long long n_ = s_blk[w]; s_blk[w] = total; total += n_; }
          s_blk[32] = total ? atomicAdd(P.queue_head, total) : (unsigned long long)P.total_units;
        }
        __syncthreads();
        const unsigned long long u = s_blk[32] + s_blk[wave] + my_rank;
        if (!alive && has_slot && u < (unsigned long long)P.total_units) stream_unit = (uint32_t)u;
        __syncthreads();                                  // (s_blk is reused below)
      }
    }
    if ((!STREAM && need != 0ull && !(queue_empty && chunk_next == chunk_end)) || (STREAM && PHASE != 1 && pass == 1)) {
      const uint32_t want = (uint32_t)__popcll(need);
      const uint32_t my_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
      uint32_t given = 0, my_unit = stream_unit;
      while (!STREAM && given < want) {
        if (chunk_next == chunk_end) {
          if (queue_empty) break;
          unsigned long long start = 0;
          const uint32_t grab = P.chunk;
          if (lane == 0) {
            start = atomicAdd(P.queue_head, (unsigned long long)grab);
          }
          start = __shfl(start, 0);
          if (start >= P.total_units) { queue_empty = true; break; }
          chunk_next = (uint32_t)start;
          chunk_end = (uint32_t)(start + grab < P.total_units ? start + grab : P.total_units);
        }
        const uint32_t avail = chunk_end - chunk_next;
        const uint32_t take = avail < want - given ? avail : want - given;
        if (!alive && my_rank >= given && my_rank < given + take) my_unit = chunk_next + (my_rank - given);
        chunk_next += take;
        given += take;
      }
      if (my_unit != kMissTri) {
        uint32_t x, y;
        // queue order: first every pixel's full triples, then the single-sample units (short work items last, so
        // the end of the launch does not wait for a lane that drew three long paths)
        const uint32_t g3 = opq(P.groups3), g1 = opq(P.singles), n3 = opq(P.units3), img_w = opq(S.w), img_h = opq(S.h);
        uint32_t u_pixel, u_first, u_count;
        if (my_unit < n3) { u_pixel = my_unit / g3; u_first = (my_unit % g3) * (uint32_t)NR; u_count = (uint32_t)NR; }
        else { const uint32_t v = my_unit - n3; u_pixel = v / g1; u_first = g3 * (uint32_t)NR + v % g1; u_count = 1u; }
        unit_pixel(opq(P.T), u_pixel, x, y);
        if (x < img_w && y < img_h) {                   // padding pixels of edge tiles are never read
          alive = true;
          burst = true;
          pxy = x | (y << 16);
          pixel_slot = u_pixel;
          samp = u_first | (u_first << 16) | (u_count << 30);
          level = 0;
          depth = S.max_depth;
          // camera rays of the unit's samples (trace_pixel, student/pathtracer.cpp:26-31); absent samples repeat the first
          const Camera cam_c = opq(S.cam);
          const uint64_t seed = opq(P.seed);
          const uint32_t sample_base = opq(P.sample_base);
#pragma unroll
          for (int j = 0; j < NR; j++) {
            const uint32_t sj = u_first + ((uint32_t)j < u_count ? (uint32_t)j : 0u);
            rng.key(seed, y * img_w + x, sample_base + sj);
            const float jx = rng.unit() * 1.0f;
            const float jy = rng.unit() * 1.0f;
            const Ray cam = camera_ray(cam_c, ((float)x + jx) / (float)img_w, ((float)y + jy) / (float)img_h);
            org = cam.o;                                // the same for every camera ray (iview * origin)
            d[j] = cam.d; cb0 = cam.b0; cb1 = cam.b1;
          }
          actA = u_count > 1;                           // slot usage of a burst: ray j exists iff j < count
          actB = u_count > 2;
          need_begin = true;
          refilled = true;
        } else {
          units_finished++;                             // a padding pixel of an edge tile: nothing to render
          retry = true;                                 // (the slot is idle but NOT for lack of units: it stays on the alive list)
        }
      }
    }
    if constexpr (STREAM) {
      if (pass == 1 && TRAV == 3) {
        // ---- the next batch's rays to their fixed queue positions [batch slot][path slot]; pt_compact_kernel makes the
        // dense list the ray-cast kernel pulls from (no atomics here) ----
        bool eact[NR];
        float eb1[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) { eact[r] = alive; eb1[r] = cb1; }
        if (DL && sh_phase) {
          if (NR > 1) eact[1] = alive && sa1;
          if (NR > 2) eact[NR - 1] = alive && sa2;
#pragma unroll
          for (int r = 0; r < NR; r++) eb1[r] = sb1[r];
        } else if (burst) { if (NR > 1) eact[1] = alive && actA; if (NR > 2) eact[NR - 1] = alive && actB; }
        else if (NR == 3) { eact[0] = alive && actA; eact[1] = alive && actB; }
#pragma unroll
        for (int r = 0; r < NR; r++) {
          if (eact[r]) {
            const size_t i = (size_t)r * P.nlanes + lane_global;
            nt_store_ray(P.ray_o, 2 * i, org.x, org.y, org.z, cb0);
            nt_store_ray(P.ray_d, 2 * i, d[r].x, d[r].y, d[r].z, eb1[r]);
            emit_mask |= 1u << r;
          }
        }
        save_state();
        break;
      }
    } else {
      if (__ballot(alive) == 0ull) {
        if (queue_empty && chunk_next == chunk_end) break;
        continue;
      }
    }
    SECTION_END(ST_REFILL)

    if constexpr (TRAV == 4 && PHASE == 2) {
      // ---- the probe kernel's whole trace step: which rays of the batch can reach a queued mesh?  A ray needs the walk of mesh m
      // iff it hits either child box of the node above m's leaf (the fused kernel's probe pass computes exactly this from the
      // top-down sweep's flags: BBox::hit's verdict does not depend on the `times` handed down, only their narrowing does).
      bool pact[NR];
#pragma unroll
      for (int r = 0; r < NR; r++) pact[r] = alive;
      if (burst) { if (NR > 1) pact[1] = alive && actA; if (NR > 2) pact[NR - 1] = alive && actB; }
      else if (NR == 3) { pact[0] = alive && actA; pact[1] = alive && actB; }
      float pb0[NR], pb1[NR];
#pragma unroll
      for (int r = 0; r < NR; r++) { pb0[r] = cb0; pb1[r] = cb1; }
      auto is_lazy = [&](uint32_t k) { const Object& ob = S.objects[k]; return ob.kind == OBJ_MESH && ob.use_bvh != 0u && ob.nrec > 0u; };
      bool h_[NR]; float dd_[NR]; uint32_t tt_[NR];
      if (Q == 0) {
        for (uint32_t k = 0; k < nobj; k++)
          if (is_lazy(k)) object_queueN<NR>(S, k, org, d, pb0, pb1, h_, dd_, tt_, pact, P, lane_global, true, emit_mask);
      } else {
        V3 pinv[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) pinv[r] = v3(1.0f / d[r].x, 1.0f / d[r].y, 1.0f / d[r].z);
        for (uint32_t q = 0; q < Q; q++) {
          const WaveInterior& W = S.wave_tlas[q];
          const bool lazy_l = W.l_ref < 0 && W.l_cnt == 1u && is_lazy((uint32_t)~W.l_ref);
          const bool lazy_r = W.r_ref < 0 && W.r_cnt == 1u && is_lazy((uint32_t)~W.r_ref);
          if (!lazy_l && !lazy_r) continue;
          bool pneed[NR];
#pragma unroll
          for (int r = 0; r < NR; r++) {
            float ta = 0.0f, tb = 0.0f, tc = 0.0f, td = 0.0f;
            const bool hl = box_hit_inv(W.boxl, org, pinv[r], ta, tb);
            const bool hr = box_hit_inv(W.boxr, org, pinv[r], tc, td);
            pneed[r] = pact[r] && (hl || hr);
          }
          if (lazy_l) object_queueN<NR>(S, (uint32_t)~W.l_ref, org, d, pb0, pb1, h_, dd_, tt_, pneed, P, lane_global, true, emit_mask);
          if (lazy_r) object_queueN<NR>(S, (uint32_t)~W.r_ref, org, d, pb0, pb1, h_, dd_, tt_, pneed, P, lane_global, true, emit_mask);
        }
      }
      save_state();
      break;
    }
    // ---------------- 2. trace the batch: scene.hit for slots A, B, C ----------------
    if (TRAV != 2 && !(TRAV == 4 && pass == 1)) {
      if (DL && sh_phase) cnt.v[C_RAYS] += alive ? (1u + (sa1 ? 1u : 0u) + (sa2 ? 1u : 0u)) : 0u;
      else if (NR == 3) cnt.v[C_RAYS] += alive ? (1u + (actA ? 1u : 0u) + (actB ? 1u : 0u)) : 0u;
      else {
        // a burst traces its samples' camera rays; a bounce traces two rays where the reference issues three (continuous) or two
        cnt.v[C_RAYS] += alive ? (burst ? (1u + (actA ? 1u : 0u)) : (discrete ? 2u : 3u)) : 0u;
        traced += alive ? (burst ? (1u + (actA ? 1u : 0u)) : 2u) : 0u;
      }
    }
    const bool shb = DL && sh_phase;
    float rb0[NR], rb1[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) { rb0[r] = cb0; rb1[r] = cb1; }
    if constexpr (DL) {
#pragma unroll
      for (int r = 0; r < NR; r++) rb1[r] = shb ? sb1[r] : cb1;
    }
    Hit res[NR];
    bool act[NR];                                        // slots of this lane's batch that carry a ray
#pragma unroll
    for (int r = 0; r < NR; r++) act[r] = alive;
    if (shb) { if (NR > 1) act[1] = alive && sa1; if (NR > 2) act[NR - 1] = alive && sa2; }
    else if (burst) { if (NR > 1) act[1] = alive && actA; if (NR > 2) act[NR - 1] = alive && actB; }
    else if (NR == 3) { act[0] = alive && actA; act[1] = alive && actB; }
    bool batch_ready = alive;                            // the lane's batch has been traced completely
    if constexpr (TRAV == 3) {
      // the cast kernel traced the batch: {distance, packed ids} per slot
#pragma unroll
      for (int r = 0; r < NR; r++) {
        res[r] = no_hit();
        if (act[r]) {
          const uint2 hv = nt_load_hit(P.hits, (size_t)r * P.nlanes + lane_global);
          res[r] = unpack_ret(__uint_as_float(hv.x), hv.y, oshift);
        }
      }
    } else if constexpr (TRAV == 2) {
      // general scenes: one flattened per-lane walk over both tree levels for the slots that carry a ray
      if (need_begin) {
        cnt.v[C_RAYS] += 1u + (actA ? 1u : 0u) + (actB ? 1u : 0u);
        flat_begin(FS, S, org, d[0], d[1], d[2], cb0, cb1, burst || actA, burst ? actA : actB, burst ? actB : true);
        need_begin = false;
      }
      flat_run(FS, fstack, S, org, d[0], d[1], d[2], cb0, cb1, alive, P.flat_ready, P.flat_interior);
      batch_ready = alive && FS.mode == FM_DONE;
      res[0] = FS.res0; res[1] = FS.res1; res[2] = FS.res2;
      SECTION_END(ST_LEAVES)
    } else if (Q == 0) {
      // List<Object>::hit, or a BVH<Object> whose root is a leaf: ordered fold over every object
#pragma unroll
      for (int r = 0; r < NR; r++) res[r] = no_hit();
      for (uint32_t k = 0; k < nobj; k++) {
        bool h[NR]; float dd[NR]; uint32_t tt[NR];
        object_testN<LAZY, NR, TRAV == 4>(S, k, org, d, rb0, rb1, cnt, h, dd, tt, act, cidx, &P, lane_global, pass == 1, &emit_mask);
#pragma unroll
        for (int r = 0; r < NR; r++) fold(res[r], h[r], dd[r], k, tt[r]);
      }
    } else {
      V3 inv[NR];
      float tx0[NR], ty0[NR];
      unsigned long long fl[NR];
      float farx0[NR];
#pragma unroll
      for (int r = 0; r < NR; r++) {
        fl[r] = 0ull; farx0[r] = 0.0f;
        inv[r] = v3(1.0f / d[r].x, 1.0f / d[r].y, 1.0f / d[r].z);
        const float dn = norm(d[r]);
        tx0[r] = rb0[r] / dn; ty0[r] = rb1[r] / dn;   // Vec2 time_initial = dist_bounds / dir.norm()
      }
      // top-down: box tests, nearer/farther decision, hand `times` down
      for (uint32_t q = 0; q < Q; q++) {
        const WaveInterior& W = S.wave_tlas[q];
#pragma unroll
        for (int r = 0; r < NR; r++) {
          float tx = tx0[r], ty = ty0[r];
          if (q != 0) { tx = SLOT(q, r, 0); ty = SLOT(q, r, 1); }
          float t1x = tx, t1y = ty, t2x = tx, t2y = ty;
          const bool hl = box_hit_inv(W.boxl, org, inv[r], t1x, t1y);
          const bool hr = box_hit_inv(W.boxr, org, inv[r], t2x, t2y);
          // closer/second as in student/bvh.inl:186-209: both hit -> smaller entry time first (ties: right);
          // one hit -> that one, the other child gets ray.dist_bounds as its `times`
          const bool hb = hl && hr;
          const bool cl = hb ? (t1x < t2x) : hl;
          const float cx = cl ? t1x : t2x, cy = cl ? t1y : t2y;
          const float fx = hb ? (cl ? t2x : t1x) : rb0[r];
          const float fy = hb ? (cl ? t2y : t1y) : rb1[r];
          fl[r] |= (unsigned long long)((hl ? 1u : 0u) | (hr ? 2u : 0u) | (cl ? 4u : 0u) | (hb ? 8u : 0u)) << (4 * q);
          if (q != 0) SLOT(q, r, 0) = fx; else farx0[r] = fx;
          if (W.l_ref >= 0) { SLOT(W.l_ref, r, 0) = cl ? cx : fx; SLOT(W.l_ref, r, 1) = cl ? cy : fy; }
          if (W.r_ref >= 0) { SLOT(W.r_ref, r, 0) = cl ? fx : cx; SLOT(W.r_ref, r, 1) = cl ? fy : cy; }
        }
      }
      SECTION_END(ST_TOPDOWN)
      // bottom-up: leaves are evaluated in place, interior children read back, then the visit rule + Trace::min
      for (int q = (int)Q - 1; q >= 0; q--) {
        const WaveInterior& W = S.wave_tlas[q];
        Hit L[NR], R[NR];
        // A leaf child that holds a mesh with a real BVH<Triangle> is evaluated after its sibling and only for the rays
        // whose traversal can reach it: find_closest_hit visits the nearer child, and the other one iff
        // `cur_far_t.x < ret.distance || (!ret.hit && hitboth)` (student/bvh.inl:216) - whether node q itself is reached is
        // not known yet (bottom-up), so that part is assumed.  Rays that do not need it get "no hit", which the
        // combination below never selects.
        auto lazy_leaf = [&](int32_t ref, uint32_t n) {
          if (!LAZY || ref >= 0 || n != 1u) return false;
          const Object& ob = S.objects[(uint32_t)~ref];
          return ob.kind == OBJ_MESH && ob.use_bvh != 0u && ob.nrec > 0u;
        };
        auto eval_child = [&](int32_t ref, uint32_t n, Hit* out, const bool* need) {
          if (ref >= 0) {
#pragma unroll
            for (int r = 0; r < NR; r++) out[r] = unpack_ret(SLOT(ref, r, 0), __float_as_uint(SLOT(ref, r, 1)));
          } else {
#pragma unroll
            for (int r = 0; r < NR; r++) out[r] = no_hit();
            const uint32_t first = (uint32_t)~ref;
            for (uint32_t k = first; k < first + n; k++) {
              bool h[NR]; float dd[NR]; uint32_t tt[NR];
              // (TRAV 4: a mesh with a real BVH<Triangle> is a leaf of its own - checked on the host - and goes through `queued`)
              object_testN<TRAV == 1, NR>(S, k, org, d, rb0, rb1, cnt, h, dd, tt, need, cidx);
#pragma unroll
              for (int r = 0; r < NR; r++) fold(out[r], h[r], dd[r], k, tt[r]);
            }
          }
        };
        auto need_of = [&](bool is_left, const Hit* other, bool other_known, bool* need) {
#pragma unroll
          for (int r = 0; r < NR; r++) {
            const uint32_t f = (uint32_t)(fl[r] >> (4 * q)) & 15u;
            const bool nearer = ((f & 4u) != 0) == is_left;
            const float farx = (q != 0) ? SLOT(q, r, 0) : farx0[r];
            const bool second = !other_known || farx < other[r].dist || (!other[r].hit && (f & 8u));
            need[r] = act[r] && (f & 3u) != 0 && (nearer || second);
          }
        };
        if constexpr (!LAZY) {                           // no per-lane walks: both children straight, in order
          eval_child(W.l_ref, W.l_cnt, L, act);
          eval_child(W.r_ref, W.r_cnt, R, act);
        } else if constexpr (TRAV == 4) {
          // The streamed sweeps: every other child exactly as above (the very code of the Cornell kernel); a leaf that holds a
          // mesh with a real BVH<Triangle> comes after its sibling and costs a ray -> object space step and a queue access.
          // The walks are queued in the probe pass, when no mesh result exists yet, and read back in the complete pass: the
          // set of rays that "need" a mesh must not depend on any mesh's result, so a sibling whose subtree holds such a mesh
          // counts as unknown (a superset again; what is not needed is never selected).
          const uint32_t lz = S.wave_lazy[q];
          const bool lazy_l = lazy_leaf(W.l_ref, W.l_cnt), lazy_r = lazy_leaf(W.r_ref, W.r_cnt);
          auto queued = [&](int32_t ref, bool is_left, const Hit* other, bool other_known, Hit* out) {
            bool need[NR], h[NR]; float dd[NR]; uint32_t tt[NR];
            need_of(is_left, other, other_known, need);
            const uint32_t k = (uint32_t)~ref;
            object_queueN<NR>(S, k, org, d, rb0, rb1, h, dd, tt, need, P, lane_global, pass == 1, emit_mask);
#pragma unroll
            for (int r = 0; r < NR; r++) { out[r] = no_hit(); fold(out[r], h[r], dd[r], k, tt[r]); }
          };
#pragma unroll
          for (int r = 0; r < NR; r++) { L[r] = no_hit(); R[r] = no_hit(); }
          if (pass == 1) {
            // The probe pass needs no result, only the requests: the sibling counts as unknown for every ray (3 % more
            // requests than with the visit rule applied to a known sibling), and nothing else of the bottom-up sweep runs
            // - the batch is swept once, in the complete pass, instead of twice.
            if (lazy_l) queued(W.l_ref, true, R, false, L);
            if (lazy_r) queued(W.r_ref, false, L, false, R);
            continue;
          }
          if (!lazy_l) eval_child(W.l_ref, W.l_cnt, L, act);
          if (!lazy_r) eval_child(W.r_ref, W.r_cnt, R, act);
          if (lazy_l) queued(W.l_ref, true, R, (lz & 2u) == 0u, L);
          if (lazy_r) queued(W.r_ref, false, L, (lz & 1u) == 0u, R);
        } else {
        const bool lazy_l = lazy_leaf(W.l_ref, W.l_cnt), lazy_r = lazy_leaf(W.r_ref, W.r_cnt);
        // TRAV 4: the walks are queued in the probe pass, when no mesh result exists yet, and read back in the complete pass:
        // the set of rays that "need" a mesh must not depend on any mesh's result, so a sibling whose subtree holds such a
        // mesh counts as unknown (a superset again; what is not needed is never selected)
        const uint32_t lz = (TRAV == 4) ? S.wave_lazy[q] : 0u;
        const bool right_first = lazy_l && !lazy_r;      // the sibling of a lazy leaf goes first
        Hit first_out[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) { first_out[r] = no_hit(); L[r] = no_hit(); R[r] = no_hit(); }
        // one call site for both children (a loop that is not unrolled): the leaf tests are most of the kernel's code
#pragma nounroll
        for (int t = 0; t < 2; t++) {
          const bool left = (t == 0) != right_first;
          const int32_t ref = left ? W.l_ref : W.r_ref;
          const uint32_t n = left ? W.l_cnt : W.r_cnt;
          bool need[NR];
          if (left ? lazy_l : lazy_r) need_of(left, first_out, t == 1 && !(TRAV == 4 && (lz & (left ? 2u : 1u)) != 0u), need);
          else {
#pragma unroll
            for (int r = 0; r < NR; r++) need[r] = act[r];
          }
          Hit out[NR];
          eval_child(ref, n, out, need);
#pragma unroll
          for (int r = 0; r < NR; r++) {
            if (t == 0) first_out[r] = out[r];
            if (left) L[r] = out[r]; else R[r] = out[r];
          }
        }
        }
        SECTION_END(ST_LEAVES)
#pragma unroll
        for (int r = 0; r < NR; r++) {
          const uint32_t f = (uint32_t)(fl[r] >> (4 * q)) & 15u;
          const bool cl = (f & 4u) != 0;
          const Hit rc = cl ? L[r] : R[r];
          const Hit rs = cl ? R[r] : L[r];
          Hit ret = no_hit();
          if (f & 3u) {
            ret = rc;
            const float farx = (q != 0) ? SLOT(q, r, 0) : farx0[r];
            if (farx < rc.dist || (!rc.hit && (f & 8u))) {       // student/bvh.inl:216
              if (!left_wins(rc.hit, rc.dist, rs.hit, rs.dist)) ret = rs.hit ? rs : no_hit();
            }
          }
          if (q > 0) { SLOT(q, r, 0) = ret.dist; SLOT(q, r, 1) = __uint_as_float(pack_ret(ret)); }
          else res[r] = ret;
        }
        SECTION_END(ST_COMBINE)
      }
    }

    if constexpr (TRAV == 4) {
      if (pass == 1) {                                   // the probe pass: the walk requests are out, the batch waits in HBM
        save_state();
        break;
      }
    }
    // ---------------- 3. finish the previous bounce / unpack the burst, then terminate or shade ----------------
    if (batch_ready) {
      uint32_t chit = kRetMiss;                         // packed closest hit that decides how the current path goes on
      bool more_shadow = false;                         // another batch goes out before the path is resolved
      bool shadow_returned = false;
      if constexpr (DL) {
        if (sh_phase) {
          shadow_returned = true;
          // a shadow batch returned: radiance += attenuation * sample.radiance for the unoccluded lights, in light order
#pragma unroll
          for (int j = 0; j < NR; j++) {
            if (light_i + (uint32_t)j < S.ndelta) {
              const LightSample ls = delta_light_sample(S.delta_lights[light_i + (uint32_t)j], org);
              if (!res[j].hit) pl = pl + att * ls.radiance;
            }
          }
          light_i += 3u;
          if (light_i < S.ndelta) more_shadow = true;
          else {
            Spec radiance = pl;                         // sample_direct_lighting, student/pathtracer.cpp:78-172
            radiance = radiance + dA_keep;
            radiance = radiance - dA_keep;
            radiance = radiance + d6_keep;
            REC_STORE(level - 1, 0, radiance.r, radiance.g, radiance.b, discrete ? 1.0f : 0.0f);
            sh_phase = false;
            chit = held_chit;
            d[C] = dC_keep;
          }
        }
      }
      if (shadow_returned) {
      } else if (burst) {
        // burst order: slot 0 = sample s_first (continues now), the other slots = the next samples (parked)
        chit = pack_ret(res[0], oshift);
#pragma unroll
        for (int j = 0; j < NR - 1; j++) { const uint32_t pr = pack_ret(res[j + 1], oshift); if (j == 0) pend0 = pr; else pend1 = pr; }
        if constexpr (DL) {
          if (S.env_type != 0u) {
            // a camera ray that leaves the scene sees the environment light (student/pathtracer.cpp:182-188): remember
            // whether evaluate(dir) is the radiance or zero, the directions are gone when the parked samples are resolved
            // (an image map is evaluated when the sample is resolved, from its regenerated camera ray)
            if (!res[0].hit && (S.env_type != 2u || d[0].y > 0.0f)) chit = kRetMissEnv;
#pragma unroll
            for (int j = 0; j < NR - 1; j++)
              if (!res[j + 1].hit && (S.env_type != 2u || d[j + 1].y > 0.0f)) { if (j == 0) pend0 = kRetMissEnv; else pend1 = kRetMissEnv; }
          }
        }
      } else if constexpr (NR == 2) {
        // slot 0 = the direct ray that matters: discrete BSDF -> the BSDF-sampled one, else the MIS one (see the head comment)
        Spec e0 = spec(0, 0, 0);
        if (res[0].hit) { const Spec e = emissive_of(S.materials[S.objects[res[0].obj].material]); if (luma(e) > 0.0f) e0 = e; }
        Spec radiance = spec(0, 0, 0);
        if (discrete) {
          const Spec direct = e0 * att;
          radiance = radiance + direct;
        } else {
          const float pdf = (pdf4 + pdf_area) / 2.0f;
          const Spec d6 = (e0 * att) * (1.0f / pdf);
          radiance = radiance + d6;                     // ((0 + direct) - direct) + d6
        }
        REC_STORE(level - 1, 0, radiance.r, radiance.g, radiance.b, discrete ? 1.0f : 0.0f);
        chit = pack_ret(res[C], oshift);
      } else {
        if (actA) {                                     // sample_direct_lighting's arithmetic (student/pathtracer.cpp:78-172)
          Spec eA = spec(0, 0, 0), eB = spec(0, 0, 0);
          if (res[0].hit) { const Spec e = emissive_of(S.materials[S.objects[res[0].obj].material]); if (luma(e) > 0.0f) eA = e; }
          else if (DL && S.env_type != 0u) eA = env_evaluate(S, d[0]);
          if (actB && res[1].hit) { const Spec e = emissive_of(S.materials[S.objects[res[1].obj].material]); if (luma(e) > 0.0f) eB = e; }
          else if (DL && actB && S.env_type != 0u) eB = env_evaluate(S, d[1]);
          Spec radiance = spec(0, 0, 0);
          if (discrete) {
            const Spec direct = eA * att;
            radiance = radiance + direct;
          } else {
            const Spec direct = (eA * att) * (1.0f / pdf4);
            const float pdf = (pdf4 + pdf_area) / 2.0f;
            const Spec d6 = (eB * att) * (1.0f / pdf);
            if (DL && S.ndelta != 0u && luma(att) != 0.0f) {
              // point_lighting comes first in the sum: hold the two terms until the shadow rays are back
              dA_keep = direct; d6_keep = d6; pl = spec(0, 0, 0);
              light_i = 0; held_chit = pack_ret(res[C], oshift); dC_keep = d[C];
              sh_phase = true; more_shadow = true;
            } else {
              radiance = radiance + direct;
              radiance = radiance - direct;
              radiance = radiance + d6;
            }
          }
          if (!(DL && sh_phase)) {
            REC_STORE(level - 1, 0, radiance.r, radiance.g, radiance.b, discrete ? 1.0f : 0.0f);
          }
        }
        chit = pack_ret(res[C], oshift);
      }
      if constexpr (DL) {
        if (more_shadow) {
          // next shadow batch: lights light_i .. light_i + 2 (Ray(hit.pos, sample.direction, {EPS_F, distance - EPS_F}))
#pragma unroll
          for (int j = 0; j < NR; j++) {
            const uint32_t li = light_i + (uint32_t)j < S.ndelta ? light_i + (uint32_t)j : light_i;
            const LightSample ls = delta_light_sample(S.delta_lights[li], org);
            d[j] = unit(ls.direction);
            sb1[j] = ls.distance - kEps;
          }
          sa1 = light_i + 1u < S.ndelta; sa2 = light_i + 2u < S.ndelta;
          if (TRAV == 2) need_begin = true;
        }
      }
      SECTION_END(ST_POST)
      // Resolve the current path; when it ends, the next parked camera hit (if any) takes over in the same cycle.
      bool need_shade = false;
      Spec e = spec(0, 0, 0);
      uint32_t mi = 0;
      for (int guard = 0; guard < 3 && !need_shade && alive && !(DL && more_shadow); guard++) {
        const bool miss_env = DL && chit == kRetMissEnv;   // a camera ray that left the scene into the environment light
        const Hit ch = unpack_ret(0.0f, miss_env ? kRetMiss : chit, oshift);
        bool terminal = !ch.hit;                         // student/pathtracer.cpp:174-218
        e = spec(0, 0, 0);
        if (miss_env && level == 0) {
          e = spec(S.env_radiance[0], S.env_radiance[1], S.env_radiance[2]);
          if (S.env_type == 3u) {                         // Env_Map::evaluate(ray.dir) of the sample's camera ray
            Rng cr;
            const uint32_t img_w = opq(S.w), img_h = opq(S.h);
            cr.key(opq(P.seed), PY_ * img_w + PX_, opq(P.sample_base) + S_CUR_);
            const float jx = cr.unit() * 1.0f;
            const float jy = cr.unit() * 1.0f;
            const Ray cam = camera_ray(opq(S.cam), ((float)PX_ + jx) / (float)img_w, ((float)PY_ + jy) / (float)img_h);
            e = env_map_evaluate(S, cam.d);
          }
        }
        if (!terminal) {
          mi = (uint32_t)S.objects[ch.obj].material;
          e = emissive_of(S.materials[mi]);
          if (luma(e) > 0.0f) terminal = true; else e = spec(0, 0, 0);
          if (depth == 0) terminal = true;
        }
        if (!terminal) { need_shade = true; break; }
        Spec L = spec(0, 0, 0);
        for (int k = (int)level - 1; k >= 0; k--) {
          const f32x4 r0 = REC_LOAD(k, 0);
          const f32x4 r1 = REC_LOAD(k, 1);
          const Spec dk = spec(r0.x, r0.y, r0.z);
          const Spec ak = spec(r1.x, r1.y, r1.z);
          Spec ind = (r0.w != 0.0f) ? (L * ak) : ((L * ak) * r1.w);
          ind = spec(0, 0, 0) + ind;
          L = dk + ind;
        }
        const Spec out = ((level == 0) ? e : spec(0, 0, 0)) + L;
        __builtin_nontemporal_store(f32x4{out.r, out.g, out.b, 0.0f}, reinterpret_cast<f32x4*>(P.sample_out) + ((size_t)S_CUR_ * P.npix + pixel_slot));
        // next sample of the unit: its camera hit is already known
        samp = (uint32_t)samp + (1u << 16);
        if (S_CUR_ < S_FIRST_ + S_COUNT_) {
          chit = pend0;
          if (NR > 2) pend0 = pend1;
          level = 0;
          depth = S.max_depth;
          burst = true;                                 // "the ray that led here was a camera ray"
        } else {
          alive = false;
          units_finished++;
        }
      }
      SECTION_END(ST_TERMINATE)
      if (need_shade && alive) {
        const Material& m = S.materials[mi];
        const Hit ch = unpack_ret(0.0f, chit, oshift);
        Ray ray;
        ray.o = org; ray.d = d[C]; ray.b0 = cb0; ray.b1 = cb1;
        if (level == 0) {
          // a camera ray: regenerate it (and the RNG position after its two jitter draws) from the sample index
          const uint32_t img_w = opq(S.w), img_h = opq(S.h);
          rng.key(opq(P.seed), PY_ * img_w + PX_, opq(P.sample_base) + S_CUR_);
          const float jx = rng.unit() * 1.0f;
          const float jy = rng.unit() * 1.0f;
          ray = camera_ray(opq(S.cam), ((float)PX_ + jx) / (float)img_w, ((float)PY_ + jy) / (float)img_h);
        }
        burst = false;
        Surface sf = surface_of(S, ch, ray);
        if (!is_sided(m.type) && dot(sf.normal, ray.d) > 0.0f) sf.normal = neg(sf.normal);
        const Frame fr = rotate_to(sf.normal);
        const V3 out_dir = unit(frame_to_local(fr, ray.o - sf.position));
        discrete = is_discrete(m.type);
        // BSDF_Lambertian::evaluate / pdf depend on out_dir only: the reference's repeated calls
        // (scatter twice, evaluate, pdf three times) all return these two values.
        Scatter s1, s2;
        if (m.type == 0) {
          s1.atten = lambert_evaluate(m, out_dir);
          pdf4 = lambert_pdf(out_dir);
          s1.dir = lambert_direction(rng);
        } else {
          s1 = scatter(m, out_dir, rng);
        }
        const V3 world_in = frame_to_world(fr, s1.dir);
        att = s1.atten;
        actA = true;
        actB = !discrete;
        V3 chosen = world_in;
        bool log_fire = false;
        if (!discrete) {
          const V3 to_light = light_sample(S, sf.position, rng);
          chosen = rng.coin(0.5f) ? world_in : to_light;
          log_fire = rng.coin(0.0005f);                 // log_ray(world_ray_task6, 5.0f) with probability 0.0005 (student/pathtracer.cpp:148)
          pdf_area = light_pdf<false>(S, sf.position, to_light, cnt);
        }
        if (m.type == 0) { s2.atten = s1.atten; s2.dir = lambert_direction(rng); }
        else s2 = scatter(m, out_dir, rng);
        const V3 world_in2 = frame_to_world(fr, s2.dir);
        REC_STORE(level, 1, s2.atten.r, s2.atten.g, s2.atten.b, discrete ? 0.0f : (1.0f / pdf4));
        level++;
        depth--;
        org = sf.position;
        if constexpr (NR == 3) {
          d[0] = unit(world_in);                       // explicit Ray(point, dir, ...) normalizes
          d[1] = unit(chosen);
        } else {
          d[0] = unit(discrete ? world_in : chosen);   // the BSDF-sampled direct ray of a continuous BSDF is not traced
        }
        d[C] = unit(world_in2);
        cb0 = kEps; cb1 = FLT_MAX;
        need_begin = true;
#ifndef SRT_AB_NOLOG                                   // (A/B builds only: what the ray log costs the loop)
        // the logged ray is the MIS direct ray just set up - Ray(hit.pos, random_in_dir): origin and unit direction are in their
        // final registers here, nothing is computed for the log that the batch does not need anyway
        if (log_fire && S.ray_log) {
          const V3 ld = d[NR == 3 ? 1 : 0];
          log_ray_event(S.ray_log, S.ray_log_cap, org.x, org.y, org.z, ld.x, ld.y, ld.z, rng.inc, level - 1u);
        }
#endif
      }
    }
    SECTION_END(ST_SHADE)
    if constexpr (PHASE == 1) {                          // the resolve kernel: one pass; the probe kernel refills and emits
      save_state();
      break;
    }
  }
  if (STAMP && lane == 0)
    for (int i = 0; i < ST_COUNT_; i++) atomicAdd(&P.stamps[i], stamp_acc[i]);
#undef SECTION_END

#undef SLOT
#undef ST
#undef ST_SET
#undef REC_AT
#undef REC_STORE
#undef REC_LOAD
#undef PX_
#undef PY_
#undef S_FIRST_
#undef S_CUR_
#undef S_COUNT_
  unsigned long long r = cnt.v[C_RAYS], rt = (NR == 3) ? cnt.v[C_RAYS] : traced;
  for (int off = 32; off > 0; off >>= 1) { r += __shfl_down(r, off); rt += __shfl_down(rt, off); }
  if constexpr (STREAM && PHASE == 2) {
    (void)units_finished;                                // (the probe kernel traces nothing: no rays to count)
  } else if constexpr (STREAM) {
    // per-block sums into the block's own words (no atomics; pt_stream_finish_kernel adds them up after the last generation)
    (void)units_finished;
    if (lane == 0) { s_blk[wave] = r; s_blk[16 + wave] = r - rt; }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long a = 0, b = 0;
      for (uint32_t w = 0; w < blockDim.x / 64u; w++) { a += s_blk[w]; b += s_blk[16 + w]; }
      if (a) P.block_counters[2 * (size_t)blockIdx.x] += a;
      if (b) P.block_counters[2 * (size_t)blockIdx.x + 1] += b;
    }
  } else if (lane == 0) {
    atomicAdd(P.ray_counter, r);
    if (NR == 2) atomicAdd(P.elided_counter, r - rt);
  }
}

// Adds the samples of each pixel in sample order with do_trace's validity filter.  A render of more than one
// launch carries (sum, count) in `running` (4 floats per pixel slot); the last launch scales by 1/count.
__global__ void pt_reduce_kernel(TileMap T, uint32_t w, uint32_t h, uint32_t samples, const float* __restrict__ sample_out,
                                 float* __restrict__ running, int first, int last, float* __restrict__ tiles_out, const uint32_t* dev_cancel) {
  if (*dev_cancel != 0u) return;                         // srt_pt_cancel: the launch was cut short, its samples are not an epoch
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= T.local_tiles * T.tile_w * T.tile_h) return;
  uint32_t x, y;
  unit_pixel(T, p, x, y);
  const uint32_t slot = tile_slot(T, p);
  float* out = tiles_out + (size_t)slot * 3;
  if (x >= w || y >= h) { if (last) { out[0] = out[1] = out[2] = 0.0f; } return; }
  Spec acc = spec(0, 0, 0);
  uint32_t sampled = 0;
  if (!first) { acc = spec(running[4 * (size_t)p], running[4 * (size_t)p + 1], running[4 * (size_t)p + 2]); sampled = __float_as_uint(running[4 * (size_t)p + 3]); }
  const uint32_t npix = T.local_tiles * T.tile_w * T.tile_h;
  const float4* src = reinterpret_cast<const float4*>(sample_out) + p;   // [sample][pixel slot]: consecutive lanes, consecutive float4
  for (uint32_t s = 0; s < samples; s++) {
    const float4 q = src[(size_t)s * npix];
    const Spec v = spec(q.x, q.y, q.z);
    if (valid(v)) { acc = acc + v; sampled++; }
  }
  if (last) {
    if (sampled > 0) acc = acc * (1.0f / sampled);
    out[0] = acc.r; out[1] = acc.g; out[2] = acc.b;
  } else {
    running[4 * (size_t)p] = acc.r; running[4 * (size_t)p + 1] = acc.g; running[4 * (size_t)p + 2] = acc.b;
    running[4 * (size_t)p + 3] = __uint_as_float(sampled);
  }
}


// Pathtracer::do_trace's per-pixel epoch mean AND Pathtracer::accumulate's running mean (rays/pathtracer.cpp:195-231) for the samples
// of ONE launch, epoch by epoch, straight from the per-sample buffer: the device renders launches of up to 64 samples per pixel
// whatever the reference's epoch size is (samples_per_epoch = max(1, n / (hw_threads * 10)), often 1), and this kernel restores
// the reference's bookkeeping - sample j of the render belongs to epoch j / spe; an epoch's mean is the sum of its valid samples
// in order times 1 / count (a zero Spectrum when none is valid); the accumulator takes s += (mean - s) * (1.0f / k) with k the
// epoch's 1-based number.  State per pixel slot p (8 floats): {acc rgb, -, sum rgb, count}: the epoch in progress is carried from
// one launch of the render to the next.  pos = samples of the render in front of this launch, total = samples of the whole render
// (its last epoch may be short), first_k = epochs the accumulator held before the render.
__global__ void pt_fold_kernel(TileMap T, uint32_t w, uint32_t h, uint32_t samples, const float* __restrict__ sample_out, uint32_t spe,
                               uint32_t pos, uint32_t total, uint32_t first_k, float* __restrict__ state, const uint32_t* dev_cancel) {
  if (*dev_cancel != 0u) return;                         // the launch was cut short (srt_pt_cancel): nothing of it is folded
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t npix = T.local_tiles * T.tile_w * T.tile_h;
  if (p >= npix) return;
  uint32_t x, y;
  unit_pixel(T, p, x, y);
  if (x >= w || y >= h) return;                          // padding pixels of edge tiles stay zero
  float4* st = reinterpret_cast<float4*>(state) + 2 * (size_t)p;
  float4 a = st[0], c = st[1];
  Spec acc = spec(a.x, a.y, a.z), sum = spec(c.x, c.y, c.z);
  uint32_t cnt = __float_as_uint(c.w);
  const float4* src = reinterpret_cast<const float4*>(sample_out) + p;   // [sample][pixel slot]
  for (uint32_t s = 0; s < samples; s++) {
    const float4 q = src[(size_t)s * npix];
    const Spec v = spec(q.x, q.y, q.z);
    if (valid(v)) { sum = sum + v; cnt++; }
    const uint32_t idx = pos + s + 1u;
    if (idx % spe == 0u || idx == total) {
      if (cnt > 0u) sum = sum * (1.0f / cnt);
      const uint32_t k = first_k + (idx + spe - 1u) / spe;
      acc = acc + (sum - acc) * (1.0f / k);
      sum = spec(0, 0, 0); cnt = 0u;
    }
  }
  st[0] = make_float4(acc.r, acc.g, acc.b, 0.0f);
  st[1] = make_float4(sum.r, sum.g, sum.b, __uint_as_float(cnt));
}

// The accumulator of this rank's pixels as tile radiance (the layout srt_pt_render_epoch_device writes and the gather moves).
__global__ void pt_acc_image_kernel(TileMap T, const float* __restrict__ state, float* __restrict__ tiles_out) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= T.local_tiles * T.tile_w * T.tile_h) return;
  const float4 a = reinterpret_cast<const float4*>(state)[2 * (size_t)p];
  float* out = tiles_out + (size_t)tile_slot(T, p) * 3;
  out[0] = a.x; out[1] = a.y; out[2] = a.z;
}

}  // namespace srt

#endif
