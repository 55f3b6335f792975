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
//   phase 1  for every object k (uniform loop; the object's matrices and triangles arrive through the
//            scalar unit, no per-lane scene loads): all 64 lanes transform their up-to-3 rays into object
//            space — the three rays of a bounce share their origin, so the origin is transformed once —
//            run Sphere::hit / the ordered Triangle::hit fold, and park (hit, world distance, triangle) in
//            a per-lane LDS table.  Only meshes with a real BVH<Triangle> fall back to a per-lane walk.
//   phase 2  each lane replays find_closest_hit over the small top-level tree (nodes staged in LDS): the
//            box tests and the visit rule are the reference's, a leaf visit is a table lookup.  Objects the
//            reference would not have visited are never looked up, so the result is identical by construction.
//
// Lanes run bounce cycles in lockstep: shade -> {BSDF-sampled direct ray, MIS direct ray, indirect ray} in
// one batch -> shade ...  A lane whose path ends pulls the next (pixel, sample) unit from a global queue
// (one atomic per 512 units per wave) and uses the indirect slot for its camera ray: persistent threads
// with per-lane path regeneration, no tail of idle lanes.  Every sample's radiance is written to a
// per-unit buffer; pt_reduce_kernel then adds the samples of a pixel in sample order, exactly as do_trace
// does (rays/pathtracer.cpp:216-226), so the image does not depend on which lane traced what.
#ifndef SRT_PT_WAVE_H
#define SRT_PT_WAVE_H

#include "pt_trace.h"

namespace srt {

constexpr uint32_t kWaveMaxObjects = 16;
constexpr uint32_t kChunk = 512;          // units a wave reserves per queue atomic
constexpr uint32_t kMissTri = 0xFFFFFFFFu;
constexpr int kRecFields = 8;             // direct rgb, atten rgb, inv_pdf, discrete

struct WaveParams {
  TileMap T;
  uint64_t seed;
  uint32_t sample_base;      // first sample index of this launch
  uint32_t samples;          // samples per pixel in this launch
  uint32_t total_units;      // local_tiles * tile_w * tile_h * samples
  uint32_t nlanes;           // threads of the whole grid (record scratch stride)
  float* sample_out;         // [unit][3]
  float* records;            // [(level * kRecFields + f) * nlanes + lane]
  unsigned long long* queue_head;   // next unit to hand out (zeroed before every launch)
  unsigned long long* ray_counter;  // scene.hit calls, accumulated across launches
};

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

__global__ __launch_bounds__(256) void pt_wave_kernel(DScene S, WaveParams P) {
  extern __shared__ float4 lds_raw[];
  Node* lnodes = reinterpret_cast<Node*>(lds_raw);
  const uint32_t node_f4 = S.tlas_nodes * 2;  // 32 B per node
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t nobj = S.nobjects;
  // per-wave tables: dist[k][r][lane], tri[k][r][lane]
  float* tdist = reinterpret_cast<float*>(lds_raw + node_f4) + (size_t)wave * nobj * 3 * 64 * 2;
  uint32_t* ttri = reinterpret_cast<uint32_t*>(tdist + nobj * 3 * 64);
  for (uint32_t i = threadIdx.x; i < node_f4; i += blockDim.x) lds_raw[i] = reinterpret_cast<const float4*>(S.nodes)[i];
  __syncthreads();

  const uint32_t lane_global = blockIdx.x * blockDim.x + threadIdx.x;
  Counters cnt;
  cnt.v[C_RAYS] = 0;

  // ---- persistent per-lane path state ----
  bool alive = false;
  uint32_t unit_id = 0, depth = 0, level = 0;
  Rng rng;
  rng.state = 0; rng.inc = 1; rng.draws = 0;
  V3 org = v3(0, 0, 0);
  V3 d[3] = {v3(0, 0, 1), v3(0, 0, 1), v3(0, 0, 1)};   // A: BSDF direct, B: MIS direct, C: indirect / camera
  float cb0 = 0.0f, cb1 = 0.0f;                          // bounds of slot C (A and B are always [EPS_F, FLT_MAX])
  bool actA = false, actB = false;
  Spec att = spec(0, 0, 0);                              // s1.attenuation (== evaluate(out) for Lambertian)
  float pdf4 = 1.0f, pdf_area = 0.0f;
  bool discrete = false;

  // wave-uniform queue window
  uint32_t chunk_next = 0, chunk_end = 0;
  bool queue_empty = false;

  for (;;) {
    // ---------------- 1. refill idle lanes ----------------
    const unsigned long long need = __ballot(!alive);
    if (need != 0ull && !(queue_empty && chunk_next == chunk_end)) {
      const uint32_t want = (uint32_t)__popcll(need);
      const uint32_t my_rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
      uint32_t given = 0, my_unit = kMissTri;
      while (given < want) {
        if (chunk_next == chunk_end) {
          if (queue_empty) break;
          unsigned long long start = 0;
          if (lane == 0) start = atomicAdd(P.queue_head, (unsigned long long)kChunk);
          start = __shfl(start, 0);
          if (start >= P.total_units) { queue_empty = true; break; }
          chunk_next = (uint32_t)start;
          chunk_end = (uint32_t)(start + kChunk < P.total_units ? start + kChunk : P.total_units);
        }
        const uint32_t avail = chunk_end - chunk_next;
        const uint32_t take = avail < want - given ? avail : want - given;
        if (!alive && my_rank >= given && my_rank < given + take) my_unit = chunk_next + (my_rank - given);
        chunk_next += take;
        given += take;
      }
      if (my_unit != kMissTri) {
        uint32_t x, y;
        unit_pixel(P.T, my_unit / P.samples, x, y);
        if (x < S.w && y < S.h) {                       // padding pixels of edge tiles are never read
          alive = true;
          unit_id = my_unit;
          level = 0;
          depth = S.max_depth;
          rng.key(P.seed, y * S.w + x, P.sample_base + my_unit % P.samples);
          const float jx = rng.unit() * 1.0f;           // trace_pixel, student/pathtracer.cpp:26-31
          const float jy = rng.unit() * 1.0f;
          const Ray cam = camera_ray(S, ((float)x + jx) / (float)S.w, ((float)y + jy) / (float)S.h);
          org = cam.o;
          d[2] = cam.d; cb0 = cam.b0; cb1 = cam.b1;
          d[0] = cam.d; d[1] = cam.d;                   // inactive slots carry a harmless copy
          actA = actB = false;
        }
      }
    }
    if (__ballot(alive) == 0ull) {
      if (queue_empty && chunk_next == chunk_end) break;
      continue;
    }

    // ---------------- 2. trace the batch ----------------
    cnt.v[C_RAYS] += alive ? (1u + (actA ? 1u : 0u) + (actB ? 1u : 0u)) : 0u;
    const float rb0[3] = {kEps, kEps, cb0};
    const float rb1[3] = {FLT_MAX, FLT_MAX, cb1};

    // phase 1: every object, wave-uniformly
    for (uint32_t k = 0; k < nobj; k++) {
      const Object& o = S.objects[k];
      const bool xf = o.has_trans != 0;
      V3 oorg = org;
      if (xf) oorg = mat_point(o.itrans, org);
      bool hit[3];
      float dist[3];
      uint32_t tri[3];
      V3 pos[3];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        Ray ray;
        ray.o = oorg; ray.d = d[r]; ray.b0 = rb0[r]; ray.b1 = rb1[r];
        if (xf) {                                       // Ray::transform with the shared origin
          ray.d = mat_rotate(o.itrans, d[r]);
          const float dn = norm(ray.d);
          ray.b0 *= dn; ray.b1 *= dn;
          ray.d = ray.d / dn;
        }
        tri[r] = 0;
        if (o.kind == OBJ_SPHERE) {
          const SphHit sh = sphere_hit(o.radius, ray);
          hit[r] = sh.hit;
          pos[r] = ray_at(ray, sh.t);
          dist[r] = fabsf(norm(pos[r] - ray.o));
        } else if (o.use_bvh && o.nnodes > 1) {         // a real BVH<Triangle>: per-lane walk
          const Hit mh = mesh_hit<false>(S, o, ray, cnt);
          hit[r] = mh.hit; dist[r] = mh.dist; tri[r] = mh.tri;
          pos[r] = v3(0, 0, 0);
          if (mh.hit && xf) { const TriHit th = tri_hit(S.tris[mh.tri], ray); pos[r] = ray_at(ray, th.t); }
        } else {                                        // <= 4 triangles in one leaf, or List<Triangle>: ordered fold
          bool bh = false; float bd = 0.0f, bt = 0.0f; uint32_t bi = 0;
          for (uint32_t t = 0; t < o.ntri; t++) {
            const TriHit th = tri_hit(S.tris[o.tri_base + t], ray);
            if (!left_wins(bh, bd, th.hit, th.dist)) {
              if (th.hit) { bh = true; bd = th.dist; bt = th.t; bi = o.tri_base + t; }
              else { bh = false; bd = 0.0f; bt = 0.0f; bi = 0; }
            }
          }
          hit[r] = bh; dist[r] = bd; tri[r] = bi;
          pos[r] = ray_at(ray, bt);
        }
      }
      if (xf && __ballot(hit[0] || hit[1] || hit[2]) != 0ull) {
        const V3 ow = mat_point(o.trans, oorg);         // Trace::transform: distance = |T*position - T*origin|
#pragma unroll
        for (int r = 0; r < 3; r++)
          if (hit[r]) dist[r] = norm(mat_point(o.trans, pos[r]) - ow);
      }
#pragma unroll
      for (int r = 0; r < 3; r++) {
        tdist[(k * 3 + r) * 64 + lane] = dist[r];
        ttri[(k * 3 + r) * 64 + lane] = hit[r] ? tri[r] : kMissTri;
      }
    }

    // phase 2: replay find_closest_hit over the top-level tree, leaves are table lookups
    Hit res[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      res[r].hit = false; res[r].dist = 0.0f; res[r].obj = 0; res[r].tri = 0;
      const bool active = alive && (r == 2 || (r == 0 ? actA : actB));
      if (__ballot(active) == 0ull) continue;
      auto leaf = [&](uint32_t slot, Hit& acc) {
        const uint32_t t = ttri[(slot * 3 + r) * 64 + lane];
        fold(acc, t != kMissTri, tdist[(slot * 3 + r) * 64 + lane], slot, t);
      };
      if (active) {
        Ray ray;
        ray.o = org; ray.d = d[r]; ray.b0 = rb0[r]; ray.b1 = rb1[r];
        if (S.use_bvh) {
          if (S.tlas_nodes) {
            const float dn = norm(ray.d);
            res[r] = traverse<kMaxTlasDepth, false>(lnodes, ray, ray.b0 / dn, ray.b1 / dn, cnt, C_TLAS, leaf);
          }
        } else {
          for (uint32_t k = 0; k < nobj; k++) leaf(k, res[r]);
        }
      }
    }

    // ---------------- 3. finish the previous bounce, then shade or terminate ----------------
    if (alive) {
      if (actA) {                                       // sample_direct_lighting's arithmetic (student/pathtracer.cpp:78-172)
        Spec eA = spec(0, 0, 0), eB = spec(0, 0, 0);
        if (res[0].hit) { const Spec e = emissive_of(S.materials[S.objects[res[0].obj].material]); if (luma(e) > 0.0f) eA = e; }
        if (actB && res[1].hit) { const Spec e = emissive_of(S.materials[S.objects[res[1].obj].material]); if (luma(e) > 0.0f) eB = e; }
        Spec radiance = spec(0, 0, 0);
        if (discrete) {
          const Spec direct = eA * att;
          radiance = radiance + direct;
        } else {
          const Spec direct = (eA * att) * (1.0f / pdf4);
          radiance = radiance + direct;
          radiance = radiance - direct;
          const float pdf = (pdf4 + pdf_area) / 2.0f;
          const Spec d6 = (eB * att) * (1.0f / pdf);
          radiance = radiance + d6;
        }
        float* rec = P.records + ((size_t)(level - 1) * kRecFields) * P.nlanes + lane_global;
        rec[0] = radiance.r; rec[(size_t)P.nlanes] = radiance.g; rec[2 * (size_t)P.nlanes] = radiance.b;
      }
      // the indirect / camera ray decides how the path goes on (student/pathtracer.cpp:174-218)
      bool terminal = !res[2].hit;
      Spec e = spec(0, 0, 0);
      uint32_t mi = 0;
      if (!terminal) {
        mi = (uint32_t)S.objects[res[2].obj].material;
        e = emissive_of(S.materials[mi]);
        if (luma(e) > 0.0f) terminal = true; else e = spec(0, 0, 0);
        if (depth == 0) terminal = true;
      }
      if (terminal) {
        Spec L = spec(0, 0, 0);
        for (int k = (int)level - 1; k >= 0; k--) {
          const float* rec = P.records + ((size_t)k * kRecFields) * P.nlanes + lane_global;
          const size_t st = P.nlanes;
          const Spec dk = spec(rec[0], rec[st], rec[2 * st]);
          const Spec ak = spec(rec[3 * st], rec[4 * st], rec[5 * st]);
          Spec ind = (rec[7 * st] != 0.0f) ? (L * ak) : ((L * ak) * rec[6 * st]);
          ind = spec(0, 0, 0) + ind;
          L = dk + ind;
        }
        const Spec out = ((level == 0) ? e : spec(0, 0, 0)) + L;
        float* so = P.sample_out + (size_t)unit_id * 3;
        so[0] = out.r; so[1] = out.g; so[2] = out.b;
        alive = false;
      } else {
        const Material& m = S.materials[mi];
        Ray ray;
        ray.o = org; ray.d = d[2]; ray.b0 = cb0; ray.b1 = cb1;
        Surface sf = surface_of(S, res[2], ray);
        if (!is_sided(m.type) && dot(sf.normal, ray.d) > 0.0f) sf.normal = neg(sf.normal);
        const Frame fr = rotate_to(sf.normal);
        const V3 out_dir = unit(frame_to_local(fr, ray.o - sf.position));
        discrete = is_discrete(m.type);
        const Scatter s1 = scatter(m, out_dir, rng);
        const V3 world_in = frame_to_world(fr, s1.dir);
        att = s1.atten;
        actA = true;
        actB = !discrete;
        V3 chosen = world_in;
        if (!discrete) {
          pdf4 = lambert_pdf(out_dir);
          const V3 to_light = light_sample(S, sf.position, rng);
          chosen = rng.coin(0.5f) ? world_in : to_light;
          (void)rng.coin(0.0005f);
          pdf_area = light_pdf<false>(S, sf.position, to_light, cnt);
        }
        const Scatter s2 = scatter(m, out_dir, rng);
        const V3 world_in2 = frame_to_world(fr, s2.dir);
        float* rec = P.records + ((size_t)level * kRecFields) * P.nlanes + lane_global;
        const size_t st = P.nlanes;
        rec[3 * st] = s2.atten.r; rec[4 * st] = s2.atten.g; rec[5 * st] = s2.atten.b;
        rec[6 * st] = discrete ? 0.0f : (1.0f / lambert_pdf(out_dir));
        rec[7 * st] = discrete ? 1.0f : 0.0f;
        level++;
        depth--;
        org = sf.position;
        d[0] = unit(world_in);                         // explicit Ray(point, dir, ...) normalizes
        d[1] = unit(chosen);
        d[2] = unit(world_in2);
        cb0 = kEps; cb1 = FLT_MAX;
      }
    }
  }

  unsigned long long r = cnt.v[C_RAYS];
  for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off);
  if (lane == 0) atomicAdd(P.ray_counter, r);
}

// Adds the samples of each pixel in sample order with do_trace's validity filter.  A render of more than one
// launch carries (sum, count) in `running` (4 floats per pixel slot); the last launch scales by 1/count.
__global__ void pt_reduce_kernel(TileMap T, uint32_t w, uint32_t h, uint32_t samples, const float* __restrict__ sample_out,
                                 float* __restrict__ running, int first, int last, float* __restrict__ tiles_out) {
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
  const float* src = sample_out + (size_t)p * samples * 3;
  for (uint32_t s = 0; s < samples; s++) {
    const Spec v = spec(src[3 * s], src[3 * s + 1], src[3 * s + 2]);
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

}  // namespace srt

#endif
