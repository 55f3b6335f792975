// Flattened per-lane scene.hit for general scenes (many objects, meshes with a real BVH<Triangle>).
//
// The reference's query is a recursion inside a recursion: BVH<Object>::find_closest_hit -> Object::hit ->
// Tri_Mesh::hit -> BVH<Triangle>::find_closest_hit -> Triangle::hit (student/bvh.inl:166-276, rays/object.h:57-65,
// student/tri_mesh.cpp:166-190).  Written as nested loops, a wave spends its time with a few lanes inside the
// innermost loop while the others wait at every loop exit (9 % of the VALU lanes were active on a 131 k-triangle
// scene, profiles/README.md).  Here the whole query is ONE loop whose iteration performs, per lane, one step of
// whatever that lane has to do next:
//   NODE    an interior record of the current tree (TLAS or a mesh's BLAS): both child boxes, nearer/farther
//           decision, push; or a leaf: fold its <= 4 triangles / start on its objects
//   OBJECT  the next object of a TLAS leaf (or of the object list): ray -> object space; sphere and single-leaf
//           meshes are finished on the spot, a BVH mesh switches the lane to that mesh's tree
//   UNWIND  one frame of the explicit stack: the visit rule for the farther child, or Trace::min of the two
//           children; leaving a mesh's tree finishes Object::hit (world distance) and folds it into the leaf
// and the three rays of a batch are walked one after the other inside the same loop, so lanes resynchronise
// once per batch, not once per ray.  Both trees are stored as 64-byte interior records (both child boxes per
// fetch) and share one stack of 16-byte frames.  The arithmetic is the one of traverse<>/traverse_records<>
// (pt_trace.h), which the parity tests pin against the reference.
#ifndef SRT_PT_FLAT_H
#define SRT_PT_FLAT_H

#include "pt_trace.h"

namespace srt {

// fl: bit 0 hitboth, bit 1 "nearer child done", bit 2 its hit flag, bits 3.. its object slot.
// Before the nearer child returns a/b hold cur_far_t; afterwards a = its distance, b = its triangle.
struct FlatFrame { int32_t second; float a; uint32_t b; uint32_t fl; };

constexpr int kFlatStack = kMaxTlasDepth + kMaxBlasDepth;
constexpr uint32_t kFlatMaxLeafObjects = 7;   // objects per BVH<Object> leaf the 3-bit count can describe (the reference builds 1)

enum : uint32_t { FM_NODE = 0, FM_OBJECT = 1, FM_UNWIND = 2, FM_DONE = 3 };

// Child reference of a TLAS record in the BLAS encoding: >= 0 interior rank, < 0 leaf with ~ref = first << 3 | count.
SRT_DEV int32_t flat_tlas_ref(int32_t ref, uint32_t cnt) {
  return ref >= 0 ? ref : ~(int32_t)((((uint32_t)~ref) << 3) | (cnt < kFlatMaxLeafObjects ? cnt : kFlatMaxLeafObjects));
}

// scene.hit for up to three rays sharing the origin `org` (slot r is traced iff act[r]); results as ids.
SRT_DEV void flat_trace3(const DScene& S, V3 org, V3 d0, V3 d1, V3 d2, float cb0, float cb1, bool act0, bool act1, bool act2,
                         Hit& res0, Hit& res1, Hit& res2) {
  FlatFrame stack[kFlatStack];
  const bool toplist = !S.use_bvh || S.wave_q == 0;   // List<Object>, or a BVH<Object> whose root is a leaf
  const uint32_t list_n = (S.use_bvh && S.tlas_nodes == 0) ? 0u : S.nobjects;

  uint32_t r = act0 ? 0u : (act1 ? 1u : (act2 ? 2u : 3u));
  uint32_t mode = FM_DONE;
  // ray of the tree the lane is in (world ray at TLAS level, object-space ray inside a mesh)
  V3 co = org, cd = d0, cinv = v3(0, 0, 0);
  float b0 = cb0, b1 = cb1, tx = 0.0f, ty = 0.0f;
  int32_t cur = 0;
  int sp = 0, base_sp = 0;
  uint32_t level = 0;                       // 0 TLAS / object list, 1 inside a mesh's BVH<Triangle>
  uint32_t obj_i = 0, obj_end = 0;          // objects of the current TLAS leaf still to do
  uint32_t rec_base = 0, tri_base = 0;      // of the mesh being walked
  bool xf = false;                          // ... and whether it has a transform
  Hit acc, ret;                             // leaf accumulator (TLAS level) / result of the subtree just finished
  acc.hit = false; acc.dist = 0.0f; acc.obj = 0; acc.tri = 0;
  ret = acc;
  res0 = acc; res1 = acc; res2 = acc;

#define SRT_FLAT_BEGIN_RAY()                                                                   \
  {                                                                                            \
    co = org; cd = (r == 0u) ? d0 : ((r == 1u) ? d1 : d2); b0 = cb0; b1 = cb1;                 \
    cinv = v3(1.0f / cd.x, 1.0f / cd.y, 1.0f / cd.z);                                          \
    level = 0; sp = 0; base_sp = 0;                                                            \
    acc.hit = false; acc.dist = 0.0f; acc.obj = 0; acc.tri = 0;                                \
    if (toplist) { obj_i = 0; obj_end = list_n; mode = FM_OBJECT; }                            \
    else {                                                                                     \
      const float dn_ = norm(cd);                                                              \
      tx = b0 / dn_; ty = b1 / dn_;            /* Vec2 times = dist_bounds / dir.norm() */      \
      cur = 0; mode = FM_NODE;                                                                 \
    }                                                                                          \
  }

  if (r < 3u) SRT_FLAT_BEGIN_RAY()

  while (__ballot(mode != FM_DONE) != 0ull) {
    // ------------------------------------------------------------------ NODE
    if (mode == FM_NODE) {
      if (cur >= 0) {
        const WaveInterior* __restrict__ rp = level ? (S.blas_recs + rec_base) : S.wave_tlas;
        const WaveInterior W = rp[cur];
        float t1x = tx, t1y = ty, t2x = tx, t2y = ty;
        const bool hl = box_hit_rec(W.boxl, co, cinv, t1x, t1y);
        const bool hr = box_hit_rec(W.boxr, co, cinv, t2x, t2y);
        if (hl || hr) {
          const int32_t lref = level ? W.l_ref : flat_tlas_ref(W.l_ref, W.l_cnt);
          const int32_t rref = level ? W.r_ref : flat_tlas_ref(W.r_ref, W.r_cnt);
          const bool hb = hl && hr;
          const bool cl = hb ? (t1x < t2x) : hl;     // both hit: smaller entry time first, ties go right
          FlatFrame f;
          f.second = cl ? rref : lref;
          f.a = hb ? (cl ? t2x : t1x) : b0;          // cur_far_t: the other child's times, or ray.dist_bounds
          f.b = __float_as_uint(hb ? (cl ? t2y : t1y) : b1);
          f.fl = hb ? 1u : 0u;
          stack[sp++] = f;
          cur = cl ? lref : rref;
          tx = cl ? t1x : t2x;
          ty = cl ? t1y : t2y;
        } else {
          ret.hit = false; ret.dist = 0.0f; ret.obj = 0; ret.tri = 0;
          mode = FM_UNWIND;
        }
      } else if (level) {                              // leaf of a BVH<Triangle>: fold its triangles in order
        const uint32_t packed = (uint32_t)~cur;
        const uint32_t first = tri_base + (packed >> 3), n = packed & 7u;
        Ray ray; ray.o = co; ray.d = cd; ray.b0 = b0; ray.b1 = b1;
        ret.hit = false; ret.dist = 0.0f; ret.obj = 0; ret.tri = 0;
        for (uint32_t i = 0; i < n; i++) {
          const TriHit th = tri_hit(S.tris[first + i], ray);
          fold(ret, th.hit, th.dist, 0, first + i);
        }
        mode = FM_UNWIND;
      } else {                                         // leaf of the BVH<Object>
        const uint32_t packed = (uint32_t)~cur;
        obj_i = packed >> 3; obj_end = obj_i + (packed & 7u);
        acc.hit = false; acc.dist = 0.0f; acc.obj = 0; acc.tri = 0;
        mode = FM_OBJECT;
      }
    }
    // ------------------------------------------------------------------ OBJECT (Object::hit, rays/object.h:57-65)
    if (mode == FM_OBJECT) {
      if (obj_i >= obj_end) {
        ret = acc;
        mode = FM_UNWIND;
      } else {
        const Object& o = S.objects[obj_i];
        Ray ray; ray.o = co; ray.d = cd; ray.b0 = b0; ray.b1 = b1;     // level 0: the world ray
        const bool oxf = o.has_trans != 0;
        if (oxf) ray_transform(ray, o.itrans);
        if (o.kind != OBJ_SPHERE && o.use_bvh && o.nrec > 0) {        // Tri_Mesh with a real BVH: walk it
          co = ray.o; cd = ray.d; b0 = ray.b0; b1 = ray.b1;
          cinv = v3(1.0f / cd.x, 1.0f / cd.y, 1.0f / cd.z);
          const float dn = norm(cd);
          tx = b0 / dn; ty = b1 / dn;
          level = 1; rec_base = o.rec_base; tri_base = o.tri_base; xf = oxf;
          base_sp = sp; cur = 0;
          mode = FM_NODE;
        } else {
          bool hit; float dist; uint32_t tri = 0; V3 pos;
          if (o.kind == OBJ_SPHERE) {
            const SphHit sh = sphere_hit(o.radius, ray);
            hit = sh.hit;
            pos = ray_at(ray, sh.t);
            dist = fabsf(norm(pos - ray.o));
          } else {                                     // one leaf / List<Triangle>: ordered fold over every triangle
            Hit best; best.hit = false; best.dist = 0.0f; best.obj = 0; best.tri = 0;
            float bt = 0.0f;
            const uint32_t n = (o.use_bvh && o.nnodes == 0) ? 0u : o.ntri;
            for (uint32_t t = 0; t < n; t++) {
              const TriHit th = tri_hit(S.tris[o.tri_base + t], ray);
              const bool keep = left_wins(best.hit, best.dist, th.hit, th.dist);
              bt = keep ? bt : (th.hit ? th.t : 0.0f);
              fold(best, th.hit, th.dist, 0, o.tri_base + t);
            }
            hit = best.hit; dist = best.dist; tri = best.tri;
            pos = ray_at(ray, bt);
          }
          if (hit && oxf) {                            // Trace::transform: distance = |T*position - T*origin|
            const V3 pw = mat_point(o.trans, pos);
            const V3 ow = mat_point(o.trans, ray.o);
            dist = norm(pw - ow);
          }
          fold(acc, hit, dist, obj_i, tri);
          obj_i++;
        }
      }
    }
    // ------------------------------------------------------------------ UNWIND
    if (mode == FM_UNWIND) {
      if (level && sp == base_sp) {                    // the mesh's tree is done: finish Object::hit, back to the leaf
        bool hit = ret.hit; float dist = ret.dist; const uint32_t tri = ret.tri;
        if (hit && xf) {
          const Object& o = S.objects[obj_i];
          Ray ray; ray.o = co; ray.d = cd; ray.b0 = b0; ray.b1 = b1;
          const TriHit th = tri_hit(S.tris[tri], ray);  // (u, v, t) of the winner: same arithmetic, same bits
          const V3 pos = ray_at(ray, th.t);
          const V3 pw = mat_point(o.trans, pos);
          const V3 ow = mat_point(o.trans, ray.o);
          dist = norm(pw - ow);
        }
        fold(acc, hit, dist, obj_i, tri);
        obj_i++;
        level = 0;
        co = org; cd = (r == 0u) ? d0 : ((r == 1u) ? d1 : d2); b0 = cb0; b1 = cb1;
        cinv = v3(1.0f / cd.x, 1.0f / cd.y, 1.0f / cd.z);
        mode = FM_OBJECT;
      } else if (sp == 0) {                            // the query of ray r is complete
        if (r == 0u) res0 = ret; else if (r == 1u) res1 = ret; else res2 = ret;
        r = (r == 0u) ? (act1 ? 1u : (act2 ? 2u : 3u)) : ((r == 1u) ? (act2 ? 2u : 3u) : 3u);
        mode = FM_DONE;
        if (r < 3u) SRT_FLAT_BEGIN_RAY()
      } else {
        const FlatFrame f = stack[sp - 1];
        const bool near_done = (f.fl & 2u) != 0;
        // student/bvh.inl:216: after the nearer child, also visit the farther one iff ...
        const bool visit = !near_done && (f.a < ret.dist || (!ret.hit && (f.fl & 1u) != 0));
        if (visit) {
          FlatFrame g;                                 // keep the nearer child's result in the frame
          g.second = f.second; g.a = ret.dist; g.b = ret.tri;
          g.fl = f.fl | 2u | (ret.hit ? 4u : 0u) | (ret.obj << 3);
          stack[sp - 1] = g;
          cur = f.second; tx = f.a; ty = __uint_as_float(f.b);
          mode = FM_NODE;
        } else {
          // nearer child only: this node's result is `ret` as it stands; after the farther child:
          // Trace::min(nearer, farther) - the nearer result wins only when strictly closer (a miss keeps ret's zeros)
          const bool saved = near_done && left_wins((f.fl & 4u) != 0, f.a, ret.hit, ret.dist);
          ret.hit = saved ? true : ret.hit;
          ret.dist = saved ? f.a : ret.dist;
          ret.obj = saved ? (f.fl >> 3) : ret.obj;
          ret.tri = saved ? f.b : ret.tri;
          sp--;
        }
      }
    }
  }
#undef SRT_FLAT_BEGIN_RAY
}

}  // namespace srt

#endif
