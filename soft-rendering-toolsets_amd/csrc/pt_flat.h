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

// Keeps a value in a VGPR of its own at this point (an empty asm the optimizer cannot look through).  The CPU build of
// the test harness (tests/host_emu) defines it away.
#ifndef SRT_PIN_VGPR
#define SRT_PIN_VGPR(x) asm volatile("" : "+v"(x))
#endif

// fl: bit 0 hitboth, bit 1 "nearer child done", bit 2 its hit flag, bits 3.. its object slot.
// Before the nearer child returns a/b hold cur_far_t; afterwards a = its distance, b = its triangle.
struct FlatFrame { int32_t second; float a; uint32_t b; uint32_t fl; };

// Where the frames live is the caller's choice: ArrayStack is a per-lane array (scratch memory on the device); the streamed
// ray-cast kernel (pt_stream.h) keeps 12-byte packed frames in LDS, [depth][word][lane], which is conflict-free however
// the lanes' depths differ.
struct ArrayStack {
  FlatFrame* p;
  SRT_DEV FlatFrame load(int i) const { return p[i]; }
  SRT_DEV void store(int i, const FlatFrame& f) const { p[i] = f; }
};

constexpr int kFlatStack = kMaxTlasDepth + kMaxBlasDepth;
constexpr uint32_t kFlatMaxLeafObjects = 7;   // objects per BVH<Object> leaf the 3-bit count can describe (the reference builds 1)

enum : uint32_t { FM_NODE = 0, FM_OBJECT = 1, FM_UNWIND = 2, FM_DONE = 3 };

// Child reference of a TLAS record in the BLAS encoding: >= 0 interior rank, < 0 leaf with ~ref = first << 3 | count.
SRT_DEV int32_t flat_tlas_ref(int32_t ref, uint32_t cnt) {
  return ref >= 0 ? ref : ~(int32_t)((((uint32_t)~ref) << 3) | (cnt < kFlatMaxLeafObjects ? cnt : kFlatMaxLeafObjects));
}

// Per-lane state of the walk.  It lives in registers across iterations of the caller's loop, so a wave may leave
// the walk while some lanes are still inside a tree (to shade the lanes that are finished) and come back later.
struct FlatState {
  uint32_t r = 3;                           // slot being traced (3: none left)
  uint32_t mode = FM_DONE;
  bool act1 = false, act2 = false;          // slots 1, 2 still to do
  // ray of the tree the lane is in (world ray at TLAS level, object-space ray inside a mesh)
  V3 co, cd, cinv;
  float b0 = 0.0f, b1 = 0.0f, tx = 0.0f, ty = 0.0f;
  int32_t cur = 0;
  int sp = 0, base_sp = 0;
  uint32_t level = 0;                       // 0 TLAS / object list, 1 inside a mesh's BVH<Triangle>
  uint32_t obj_i = 0, obj_end = 0;          // objects of the current TLAS leaf still to do
  uint32_t rec_base = 0, tri_base = 0;      // of the mesh being walked
  bool xf = false;                          // ... and whether it has a transform
  bool inv_ok = false;                      // (walk requests only) cinv is finite in every component: set once per request
  Hit acc, ret;                             // leaf accumulator (TLAS level) / result of the subtree just finished
  Hit res0, res1, res2;
};

SRT_DEV V3 flat_pick(uint32_t r, V3 d0, V3 d1, V3 d2) {
  return v3(r == 0u ? d0.x : (r == 1u ? d1.x : d2.x), r == 0u ? d0.y : (r == 1u ? d1.y : d2.y), r == 0u ? d0.z : (r == 1u ? d1.z : d2.z));
}
SRT_DEV Hit flat_select(bool c, const Hit& a, const Hit& b) {
  Hit h; h.hit = c ? a.hit : b.hit; h.dist = c ? a.dist : b.dist; h.obj = c ? a.obj : b.obj; h.tri = c ? a.tri : b.tri; return h;
}
SRT_DEV Hit flat_no_hit() { Hit h; h.hit = false; h.dist = 0.0f; h.obj = 0; h.tri = 0; return h; }

// Start on slot F.r (< 3) of the batch.
SRT_DEV void flat_begin_ray(FlatState& F, const DScene& S, V3 org, V3 d0, V3 d1, V3 d2, float cb0, float cb1) {
  const bool toplist = !S.use_bvh || S.wave_q == 0;   // List<Object>, or a BVH<Object> whose root is a leaf
  F.co = org; F.cd = flat_pick(F.r, d0, d1, d2); F.b0 = cb0; F.b1 = cb1;
  F.cinv = v3(1.0f / F.cd.x, 1.0f / F.cd.y, 1.0f / F.cd.z);
  F.level = 0; F.sp = 0; F.base_sp = 0;
  F.acc = flat_no_hit();
  if (toplist) {
    F.obj_i = 0; F.obj_end = (S.use_bvh && S.tlas_nodes == 0) ? 0u : S.nobjects; F.mode = FM_OBJECT;
  } else {
    const float dn = norm(F.cd);
    F.tx = F.b0 / dn; F.ty = F.b1 / dn;     // Vec2 times = dist_bounds / dir.norm()
    F.cur = 0; F.mode = FM_NODE;
  }
}

// Start a batch: slot r is traced iff act_r.
SRT_DEV void flat_begin(FlatState& F, const DScene& S, V3 org, V3 d0, V3 d1, V3 d2, float cb0, float cb1, bool act0, bool act1,
                        bool act2) {
  F.res0 = flat_no_hit(); F.res1 = F.res0; F.res2 = F.res0; F.ret = F.res0;
  F.act1 = act1; F.act2 = act2;
  F.r = act0 ? 0u : (act1 ? 1u : (act2 ? 2u : 3u));
  F.mode = FM_DONE;
  if (F.r < 3u) flat_begin_ray(F, S, org, d0, d1, d2, cb0, cb1);
}

// ---- the steps of the walk; each requires the state named in its comment ----

// mode == NODE, cur >= 0: an interior record of the current tree.
// NO FRAME when only one child box is hit, the other child is an interior node and the ray's reciprocal direction is
// finite.  The reference would come back to that other child iff the hit child's subtree reports a hit (cur_far_t =
// dist_bounds, student/bvh.inl:205-216) and test the other child's two boxes; they lie inside a box the ray's line has just
// missed (a node's box is the exact min/max union of its primitives' boxes, so child boxes nest bit for bit), and with
// finite reciprocals every slab product of BBox::hit is monotone in the box bounds (fp subtraction and multiplication by a
// fixed finite factor are monotone; no 0 * inf), so both tests fail as well, find_closest_hit returns "no hit" there, and
// Trace::min(hit, no hit) keeps the hit: the node's result IS the hit child's.  A leaf on the other side is different -
// its primitives would be tested whatever their box says - and keeps its frame.  Saves the push, the pop and the visit.
// (LEVEL: -1 the lane's own F.level; 1 the caller knows that every lane is inside a BVH<Triangle>.)
template <typename StackT, int LEVEL = -1>
SRT_DEV void flat_interior(FlatState& F, const StackT& stack, const DScene& S) {
  const bool blas = LEVEL == 1 || (LEVEL < 0 && F.level != 0u);
  const WaveInterior* __restrict__ rp = blas ? (S.blas_recs + F.rec_base) : S.wave_tlas;
  const WaveInterior W = rp[F.cur];
  float t1x = F.tx, t1y = F.ty, t2x = F.tx, t2y = F.ty;
  const bool hl = box_hit_rec(W.boxl, F.co, F.cinv, t1x, t1y);
  const bool hr = box_hit_rec(W.boxr, F.co, F.cinv, t2x, t2y);
  if (hl || hr) {
    const int32_t lref = blas ? W.l_ref : flat_tlas_ref(W.l_ref, W.l_cnt);
    const int32_t rref = blas ? W.r_ref : flat_tlas_ref(W.r_ref, W.r_cnt);
    const bool hb = hl && hr;
    const bool cl = hb ? (t1x < t2x) : hl;     // both hit: smaller entry time first, ties go right
    const int32_t far_ref = cl ? rref : lref;
    const bool inv_finite = LEVEL == 1 ? F.inv_ok : (finite_f(F.cinv.x) && finite_f(F.cinv.y) && finite_f(F.cinv.z));
    if (hb || far_ref < 0 || !inv_finite) {
      FlatFrame f;
      f.second = far_ref;
      f.a = hb ? (cl ? t2x : t1x) : F.b0;      // cur_far_t: the other child's times, or ray.dist_bounds
      f.b = __float_as_uint(hb ? (cl ? t2y : t1y) : F.b1);
      f.fl = hb ? 1u : 0u;
      stack.store(F.sp++, f);
    }
    F.cur = cl ? lref : rref;
    F.tx = cl ? t1x : t2x;
    F.ty = cl ? t1y : t2y;
  } else {
    F.ret = flat_no_hit();
    F.mode = FM_UNWIND;
  }
}

// mode == NODE, cur < 0: a leaf.
SRT_DEV void flat_leaf(FlatState& F, const DScene& S) {
  const uint32_t packed = (uint32_t)~F.cur;
  if (F.level) {                                   // BVH<Triangle>: fold its triangles in order
    const uint32_t first = F.tri_base + (packed >> 3), n = packed & 7u;
    Ray ray; ray.o = F.co; ray.d = F.cd; ray.b0 = F.b0; ray.b1 = F.b1;
    F.ret = flat_no_hit();
    for (uint32_t i = 0; i < n; i++) {
      const TriHit th = tri_hit(S.tris[first + i], ray);
      fold(F.ret, th.hit, th.dist, 0, first + i);
    }
    F.mode = FM_UNWIND;
  } else {                                         // BVH<Object>: start on its objects
    F.obj_i = packed >> 3; F.obj_end = F.obj_i + (packed & 7u);
    F.acc = flat_no_hit();
    F.mode = FM_OBJECT;
  }
}

// mode == OBJECT: Object::hit (rays/object.h:57-65) of the next object of the leaf / list, or the end of the leaf.
SRT_DEV void flat_object(FlatState& F, const DScene& S) {
  if (F.obj_i >= F.obj_end) {
    F.ret = F.acc;
    F.mode = FM_UNWIND;
    return;
  }
  const Object& o = S.objects[F.obj_i];
  Ray ray; ray.o = F.co; ray.d = F.cd; ray.b0 = F.b0; ray.b1 = F.b1;     // level 0: the world ray
  const bool oxf = o.has_trans != 0;
  if (oxf) ray_transform(ray, o.itrans);
  if (o.kind != OBJ_SPHERE && o.use_bvh && o.nrec > 0) {        // Tri_Mesh with a real BVH: walk it
    F.co = ray.o; F.cd = ray.d; F.b0 = ray.b0; F.b1 = ray.b1;
    F.cinv = v3(1.0f / F.cd.x, 1.0f / F.cd.y, 1.0f / F.cd.z);
    const float dn = norm(F.cd);
    F.tx = F.b0 / dn; F.ty = F.b1 / dn;
    F.level = 1; F.rec_base = o.rec_base; F.tri_base = o.tri_base; F.xf = oxf;
    F.base_sp = F.sp; F.cur = 0;
    F.mode = FM_NODE;
    return;
  }
  bool hit; float dist; uint32_t tri = 0; V3 pos;
  if (o.kind == OBJ_SPHERE) {
    const SphHit sh = sphere_hit(o.radius, ray);
    hit = sh.hit;
    pos = ray_at(ray, sh.t);
    dist = fabsf(norm(pos - ray.o));
  } else {                                         // one leaf / List<Triangle>: ordered fold over every triangle
    Hit best = flat_no_hit();
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
  if (hit && oxf) {                                // Trace::transform: distance = |T*position - T*origin|
    const V3 pw = mat_point(o.trans, pos);
    const V3 ow = mat_point(o.trans, ray.o);
    dist = norm(pw - ow);
  }
  fold(F.acc, hit, dist, F.obj_i, tri);
  F.obj_i++;
}

// mode == UNWIND: is the next thing to do an ordinary frame (as opposed to leaving a mesh or finishing the ray)?
SRT_DEV bool flat_plain_frame(const FlatState& F) { return F.sp != 0 && !(F.level && F.sp == F.base_sp); }

// mode == UNWIND with an ordinary frame on top.
template <typename StackT>
SRT_DEV void flat_pop(FlatState& F, const StackT& stack) {
  const FlatFrame f = stack.load(F.sp - 1);
  // hipcc 7.2 -O2/-O3 (gfx950) has been seen to drop the `cur = f.second` below when this function is inlined
  // behind flat_interior (the lane then re-walks the nearer child instead of the farther one; -O1 and the host
  // build are right, step traces compared in round 1).  Pinning the loaded value in its own VGPR avoids it; the
  // GPU parity tests (hit() through this walk vs the nested form, and every epoch test) guard against a return.
  int32_t second = f.second;
  SRT_PIN_VGPR(second);
  const bool near_done = (f.fl & 2u) != 0;
  // student/bvh.inl:216: after the nearer child, also visit the farther one iff ...
  const bool visit = !near_done && (f.a < F.ret.dist || (!F.ret.hit && (f.fl & 1u) != 0));
  if (visit) {
    FlatFrame g;                                   // keep the nearer child's result in the frame
    g.second = second; g.a = F.ret.dist; g.b = F.ret.tri;
    g.fl = f.fl | 2u | (F.ret.hit ? 4u : 0u) | (F.ret.obj << 3);
    stack.store(F.sp - 1, g);
    F.cur = second; F.tx = f.a; F.ty = __uint_as_float(f.b);
    F.mode = FM_NODE;
  } else {
    // nearer child only: this node's result is `ret` as it stands; after the farther child:
    // Trace::min(nearer, farther) - the nearer result wins only when strictly closer (a miss keeps ret's zeros)
    const bool saved = near_done && left_wins((f.fl & 4u) != 0, f.a, F.ret.hit, F.ret.dist);
    F.ret.hit = saved ? true : F.ret.hit;
    F.ret.dist = saved ? f.a : F.ret.dist;
    F.ret.obj = saved ? (f.fl >> 3) : F.ret.obj;
    F.ret.tri = saved ? f.b : F.ret.tri;
    F.sp--;
  }
}

// mode == UNWIND without an ordinary frame: the mesh's tree is done (finish Object::hit, back to the leaf's
// objects), or the whole query of slot r is (store it, start the next slot).
SRT_DEV void flat_exit(FlatState& F, const DScene& S, V3 org, V3 d0, V3 d1, V3 d2, float cb0, float cb1) {
  if (F.level) {
    bool hit = F.ret.hit; float dist = F.ret.dist; const uint32_t tri = F.ret.tri;
    if (hit && F.xf) {
      const Object& o = S.objects[F.obj_i];
      Ray ray; ray.o = F.co; ray.d = F.cd; ray.b0 = F.b0; ray.b1 = F.b1;
      const TriHit th = tri_hit(S.tris[tri], ray);  // (u, v, t) of the winner: same arithmetic, same bits
      const V3 pos = ray_at(ray, th.t);
      const V3 pw = mat_point(o.trans, pos);
      const V3 ow = mat_point(o.trans, ray.o);
      dist = norm(pw - ow);
    }
    fold(F.acc, hit, dist, F.obj_i, tri);
    F.obj_i++;
    F.level = 0;
    F.co = org; F.cd = flat_pick(F.r, d0, d1, d2); F.b0 = cb0; F.b1 = cb1;
    F.cinv = v3(1.0f / F.cd.x, 1.0f / F.cd.y, 1.0f / F.cd.z);
    F.mode = FM_OBJECT;
  } else {
    // (value selects, not conditional stores: a store through a selected address would pin the whole state in memory)
    F.res0 = flat_select(F.r == 0u, F.ret, F.res0);
    F.res1 = flat_select(F.r == 1u, F.ret, F.res1);
    F.res2 = flat_select(F.r == 2u, F.ret, F.res2);
    F.r = (F.r == 0u) ? (F.act1 ? 1u : (F.act2 ? 2u : 3u)) : ((F.r == 1u) ? (F.act2 ? 2u : 3u) : 3u);
    F.mode = FM_DONE;
    if (F.r < 3u) flat_begin_ray(F, S, org, d0, d1, d2, cb0, cb1);
  }
}

// The walk of a whole wave, "while-while" style: interior steps - the bulk of the work, and uniform in cost - are
// repeated as long as at least `interior_min` lanes take part (lanes that reached a leaf or an object wait);
// then the waiting lanes do their leaf / object / exit step together.  Returns when no lane is walking any more, or
// when `ready_min` lanes with `counts` set have finished their batch (the caller then shades / refills those and
// calls again; the other lanes keep their place in the tree).
SRT_DEV void flat_run(FlatState& F, FlatFrame* frames, const DScene& S, V3 org, V3 d0, V3 d1, V3 d2, float cb0, float cb1,
                      bool counts, uint32_t ready_min, uint32_t interior_min) {
  const ArrayStack stack{frames};
  for (;;) {
    if (__ballot(F.mode != FM_DONE) == 0ull) break;
    if ((uint32_t)__popcll(__ballot(counts && F.mode == FM_DONE)) >= ready_min) break;
    for (;;) {
      const bool can = F.mode == FM_NODE && F.cur >= 0;
      if (__ballot(can) == 0ull) break;
      if (can) {
        flat_interior(F, stack, S);
        while (F.mode == FM_UNWIND && flat_plain_frame(F)) flat_pop(F, stack);   // a double miss: straight back to a node
      }
      if ((uint32_t)__popcll(__ballot(F.mode == FM_NODE && F.cur >= 0)) < interior_min) break;
    }
    if (F.mode == FM_NODE && F.cur < 0) flat_leaf(F, S);
    if (F.mode == FM_OBJECT) flat_object(F, S);
    while (F.mode == FM_UNWIND) {
      if (flat_plain_frame(F)) flat_pop(F, stack);
      else flat_exit(F, S, org, d0, d1, d2, cb0, cb1);
    }
  }
}

constexpr uint32_t kFlatInteriorMin = 16;

// scene.hit for up to three rays sharing the origin `org` (slot r is traced iff act_r), run to completion.
SRT_DEV void flat_trace3(const DScene& S, V3 org, V3 d0, V3 d1, V3 d2, float cb0, float cb1, bool act0, bool act1, bool act2,
                         Hit& res0, Hit& res1, Hit& res2) {
  FlatFrame stack[kFlatStack];
  FlatState F;
  flat_begin(F, S, org, d0, d1, d2, cb0, cb1, act0, act1, act2);
  flat_run(F, stack, S, org, d0, d1, d2, cb0, cb1, false, 65u, kFlatInteriorMin);
  res0 = F.res0; res1 = F.res1; res2 = F.res2;
}

}  // namespace srt

#endif
