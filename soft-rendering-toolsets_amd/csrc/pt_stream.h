// Streamed ("wavefront") form of the path tracer for scenes whose rays differ wildly in cost: meshes with a real
// BVH<Triangle> (BASELINE configs[4]: a ~100 k-triangle mesh in the Cornell box) and BVH<Object>s of any size.
//
// Why.  Inside one persistent kernel a batch of rays is a barrier: the wave waits until the longest walk of the batch has
// ended (on the 131 072-triangle scene a walk visits 1 .. 80 nodes, 14 on average; 30 % of the VALU lanes were active and
// the texture-address unit spent ~25 cycles on vector loads that carried 8 lanes, profiles/r02_cfg5_wave_base_*).  Here the
// two halves of the renderer are separate kernels that hand each other dense arrays in HBM, once per generation:
//
//   pt_wave_kernel<.., TRAV = 3, ..>  ("logic", pt_wave.h) one lane per PATH SLOT: reads the hits of the slot's batch of
//       rays, finishes sample_direct_lighting, terminates or shades (the very code of the persistent kernel), pulls a
//       new work unit when the slot's unit is finished, and appends the next batch's rays - ballot-compacted per wave,
//       one atomic per wave - to the ray queue.  Path state lives in HBM as [word][slot] planes (coalesced both ways).
//   pt_cast_kernel                      scene.hit for the queue: persistent waves, every lane walks ONE ray through both
//       tree levels (the flattened walk of pt_flat.h: same records, same visit rule, same arithmetic) and pulls the
//       next ray from the queue as soon as enough lanes of its wave are idle - a ray's length no longer holds anybody
//       up.  Traversal stacks are 12-byte frames in LDS, [depth][word][lane]: conflict-free for any mix of depths, and
//       off the vector-memory path that the record fetches saturate.
//
// A slot processes exactly one batch per generation, so the number of generations of a launch is bounded by list
// scheduling: ceil(units * M / slots) + M + 1 with M = the largest number of batches a unit can need; the host enqueues
// that many (logic, cast) pairs and every kernel returns at once when all units are done - no host synchronisation
// inside an epoch.  Results are bit-identical to every other kernel mode (tests/test_pt_gpu.py).
#ifndef SRT_PT_STREAM_H
#define SRT_PT_STREAM_H

#include "pt_flat.h"

namespace srt {

// Device words shared by the kernels of one streamed launch (zeroed before generation 0).
struct StreamCounters {
  unsigned long long queue_head;   // next work unit (refill)
  uint32_t nrays[2];               // entries of generation g's dense ray list: [g & 1]
  uint32_t cast_head[2];           // next entry of generation g to hand to a cast wave: [g & 1]
  uint32_t alive[2];               // generation g left at least one slot alive: [g & 1]
  uint32_t done;                   // set once a generation left no slot alive and the unit queue is drained
  uint32_t pad;
  uint32_t alive_n[2];             // entries of the alive-slot list generation g works from: [g & 1] (written by generation g - 1's compaction)
};

// Words of a path slot's saved state (planes of `nlanes` words).  The DL build appends its shadow-phase state.
enum {
  SW_FLAGS = 0, SW_EMIT, SW_PX, SW_PY, SW_SAMPLES, SW_PIXEL_SLOT, SW_PEND0, SW_PEND1, SW_RNG_LO, SW_RNG_HI, SW_ORG, SW_DC = SW_ORG + 3,
  SW_CB0 = SW_DC + 3, SW_CB1, SW_ATT, SW_PDF4 = SW_ATT + 3, SW_PDF_AREA, SW_BASE_WORDS,
  // DL only
  SW_D0 = SW_BASE_WORDS, SW_D1 = SW_D0 + 3, SW_LIGHT_I = SW_D1 + 3, SW_HELD, SW_PL, SW_DA = SW_PL + 3, SW_D6 = SW_DA + 3,
  SW_DCK = SW_D6 + 3, SW_SB1 = SW_DCK + 3, SW_DL_WORDS = SW_SB1 + 3
};

// 12-byte frames in LDS.  Word 0 holds the farther child's reference and "hitboth" until the nearer child has returned, then
// the nearer child's object slot and hit flag (the reference is dead by then): payload << 2 | near_done << 1 | flag.
// References: interior rank < 2^29, leaf ~(first << 3 | count) > -2^29 (checked on the host).
// The first `k` frames of a lane live in LDS, deeper ones in a per-lane column of global memory (`g`, frame i word j at
// g[((i - k) * 3 + j) * gstride]): with the childless-frame rule of flat_interior a walk rarely gets that deep, and LDS sized
// for the tree's full depth is what limited the kernel's waves per CU.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
// The streamed forms' queue traffic - rays, list entries, hits, saved path state - passes through memory once per generation:
// non-temporal accesses, so that it does not push the scene out of L2.
SRT_DEV void nt_store_ray(float4* plane, size_t i, float x, float y, float z, float w) { __builtin_nontemporal_store(f32x4{x, y, z, w}, reinterpret_cast<f32x4*>(plane) + i); }
SRT_DEV uint2 nt_load_hit(const uint2* hits, size_t i) { const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(hits) + i); return make_uint2(v.x, v.y); }
typedef __attribute__((address_space(3))) uint32_t lds_u32;   // (an explicit LDS pointer: ds_read / ds_write, never FLAT)
typedef __attribute__((address_space(3))) uint8_t lds_u8;
struct LdsStack {
  lds_u32* w;    // this lane's column: word k of frame i at w[(i * 3 + k) * 64]
  uint32_t* g;
  int k;
  uint32_t gstride;
  // The LDS access is unconditional (clamped index) and the global one sits behind a wave-uniform test: with both behind
  // per-lane branches the compiler merges them into FLAT loads of a selected address, which wait on both counters.
  // (frame index x 192 dwords with the 24-bit multiply: a full 32-bit integer multiply issues at a quarter of the rate, and this one
  //  sits in every push and every pop)
  SRT_DEV void load3(int i, uint32_t& w0, uint32_t& w1, uint32_t& w2) const {
    const bool deep = i >= k;
    const int il = deep ? 0 : i;
    const lds_u32* f = w + __umul24((uint32_t)il, 192u);
    w0 = f[0]; w1 = f[64]; w2 = f[128];
    if (__ballot(deep) != 0ull) {
      if (deep) {
        const size_t at = (size_t)(i - k) * 3u * gstride;
        w0 = g[at]; w1 = g[at + gstride]; w2 = g[at + 2u * (size_t)gstride];
      }
    }
  }
  SRT_DEV void store3(int i, uint32_t w0, uint32_t w1, uint32_t w2) const {
    const bool deep = i >= k;
    if (!deep) { lds_u32* f = w + __umul24((uint32_t)i, 192u); f[0] = w0; f[64] = w1; f[128] = w2; }
    if (__ballot(deep) != 0ull) {
      if (deep) {
        const size_t at = (size_t)(i - k) * 3u * gstride;
        g[at] = w0; g[at + gstride] = w1; g[at + 2u * (size_t)gstride] = w2;
      }
    }
  }
  SRT_DEV FlatFrame load(int i) const {
    uint32_t w0, w1, w2;
    load3(i, w0, w1, w2);
    FlatFrame f;
    f.a = __uint_as_float(w1);
    f.b = w2;
    const bool near_done = (w0 & 2u) != 0;
    f.second = near_done ? 0 : ((int32_t)w0 >> 2);
    f.fl = near_done ? (2u | ((w0 & 1u) << 2) | ((w0 >> 2) << 3)) : (w0 & 1u);
    return f;
  }
  SRT_DEV void store(int i, const FlatFrame& f) const {
    const bool near_done = (f.fl & 2u) != 0;
    const uint32_t w0 = near_done ? (((f.fl >> 3) << 2) | 2u | ((f.fl >> 2) & 1u)) : (((uint32_t)f.second << 2) | (f.fl & 1u));
    store3(i, w0, __float_as_uint(f.a), f.b);
  }
};

struct CastParams {
  const float4* ray_o;       // [2 * (queue slot * nlanes + path slot)] origin, dist_bounds.x
  const float4* ray_d;       // = ray_o + 1, same index: direction, dist_bounds.y - a request is ONE 32-byte piece (round 4; two planes were two
                             // partial lines per request on the way in and on the way out)
  const uint32_t* ray_id;    // dense list of the positions that carry a ray this generation (pt_compact_kernel)
  uint2* hits;               // [same position] {distance bits, object << obj_shift | triangle, or 0xFFFFFFFF}
  uint32_t nlanes;
  uint32_t walk_nr;          // 0: the entries are world rays (scene.hit).  NR > 0: they are walk requests of the streamed sweeps -
                             // object-space rays for the BVH<Triangle> of mesh ordinal position / nlanes / NR; the result is
                             // {world distance, triangle}
  uint32_t lazy_obj[4];      // object slot of mesh ordinal m
  StreamCounters* sc;
  uint32_t total_units;
  uint32_t gen;
  const uint32_t* nrays;     // rays of this generation
  uint32_t* head;            // queue head of this generation
  uint32_t depth;            // frames a lane may need
  uint32_t lds_frames;       // ... of which this many are in LDS,
  uint32_t* spill;           // the others here: [frame - lds_frames][word][thread of the launch]            // frames per lane
  uint32_t fetch_min;        // idle lanes of a wave that trigger a fetch
  uint32_t interior_min;     // (unused by the phase scheduler; kept for experiments)
  uint32_t leaf_min;         // lanes waiting at BVH<Triangle> leaves that make the wave run the leaf phase
  uint32_t object_min;       // lanes waiting at objects that make the wave run the object phase
  uint32_t grab;             // entries a wave takes from the pool per atomic
  uint32_t own_share;        // 0..256: this many 256ths of the list are dealt out to the waves in advance, the rest is a common pool
  uint32_t pops;             // pops a lane may take per walk trip (1 or 2)
  uint32_t obj_shift;
  unsigned long long* stats;   // STATS build only: CS_* sums over all waves
  const uint32_t* dev_cancel;  // srt_pt_cancel reached the device (pt_wave.h): nothing is cast
  const float* tri_packed;     // triangle t's record without its padding: nine floats at tri_packed + 9 t (pt_scene.h)
};

// Triangle `tri` for the cast kernel, from the records without their padding (36 B: a leaf's <= 4 triangles are 144 contiguous bytes
// instead of 192).  Round 4, on BASELINE configs[4]: interior records + triangles were 9.1 MB against the XCD's 4 MB of L2 and the
// kernel's misses 42 % of the epoch's memory-side traffic; with 7.5 MB they fell from 125.6 to 86.5 GB per epoch (the kernel's time
// did not move: it is bound by instruction issue, not by these misses).  Measured and dropped: the triangles as Tri_Mesh holds them -
// three indices + shared vertices, 2.6 MB instead of 6.3, the edges subtracted on the spot, bit-identical - where the misses went UP,
// 126 -> 198 GB (155 GB with the vertices renumbered by first use in the tree's order) and the kernel from 102 to 105.5 ms: four
// small gathers per triangle are more lines asked for than one contiguous piece, and the second, dependent round trip is not hidden.
#ifndef SRT_CAST_PACKED_TRIS
#define SRT_CAST_PACKED_TRIS 1
#endif
struct __attribute__((packed, aligned(4))) TriPacked { float v[9]; };
SRT_DEV Tri cast_load_tri(const DScene& S, const CastParams& P, uint32_t tri) {
#if SRT_CAST_PACKED_TRIS
  const TriPacked t = *reinterpret_cast<const TriPacked*>(P.tri_packed + ((tri << 3) + tri));
  Tri g;
  g.p0[0] = t.v[0]; g.p0[1] = t.v[1]; g.p0[2] = t.v[2]; g.p0[3] = 0.0f;
  g.e1[0] = t.v[3]; g.e1[1] = t.v[4]; g.e1[2] = t.v[5]; g.e1[3] = 0.0f;
  g.e2[0] = t.v[6]; g.e2[1] = t.v[7]; g.e2[2] = t.v[8]; g.e2[3] = 0.0f;
  return g;
#else
  (void)P;
  return *reinterpret_cast<const Tri*>(reinterpret_cast<const char*>(S.tris) + (((tri << 1) + tri) << 4));   // (48-byte records, tri * 48 < 2^32)
#endif
}

// Dense list of the queue positions that carry a ray: the logic kernel left one mask word per path slot (bit q: queue slot
// q); a block takes kCompactChunk consecutive path slots and appends its positions with ONE atomic.
constexpr uint32_t kCompactChunk = 8192;
// `totals` (optional): {entries queued, alive slot-generations} summed over every generation of every launch - what bench.py prices the
// streamed forms' ray-state traffic with (srt_pt_stream_counters).
__global__ __launch_bounds__(1024) void pt_compact_kernel(const uint32_t* __restrict__ emit, uint32_t nlanes, uint32_t nslots,
                                                          StreamCounters* sc, uint32_t gen, uint32_t* __restrict__ ray_id,
                                                          unsigned long long* __restrict__ totals, const uint32_t* host_cancel, uint32_t* dev_cancel,
                                                          uint32_t* __restrict__ alive_list) {
  if (sc->done != 0u) return;
  // srt_pt_cancel: this kernel's first thread looks at the host's flag once per generation (one read over PCIe); a raised flag ends
  // the launch like its last generation does - `done` - and stays in dev_cancel for the launches of the epoch still to come
  if (__hip_atomic_load(dev_cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  if (blockIdx.x == 0u && threadIdx.x == 0u && __hip_atomic_load(host_cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
    atomicExch(dev_cancel, 1u);
    sc->done = 1u;
  }
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_live[16];
  __shared__ uint32_t s_base;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t first = blockIdx.x * kCompactChunk;
  constexpr uint32_t kPer = kCompactChunk / 1024u;
  uint32_t m[kPer], n = 0, live = 0;
#pragma unroll
  for (uint32_t k = 0; k < kPer; k++) {
    const uint32_t slot = first + k * 1024u + threadIdx.x;
    m[k] = slot < nlanes ? emit[slot] : 0u;
    if (m[k] & 0x80000000u) { sc->alive[gen & 1u] = 1u; live++; }     // (every writer stores the same value)
    m[k] &= 0x7fffffffu;
    n += (uint32_t)__popc(m[k]);
  }
  for (int off = 32; off > 0; off >>= 1) live += (uint32_t)__shfl_down((int)live, off);
  if (lane == 0u) s_live[wave] = live;                    // (one global atomic per block, below: a single word takes ~90 of them per microsecond)
  uint32_t incl = n;                                      // inclusive scan over the wave, then over the block's waves
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, off); if ((int)lane >= off) incl += v; }
  if (lane == 63u) s_wave[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (uint32_t w = 0; w < 16u; w++) { const uint32_t c = s_wave[w]; s_wave[w] = total; total += c; }
    s_base = total ? atomicAdd(&sc->nrays[gen & 1u], total) : 0u;
    if (totals) {
      uint32_t alive_slots = 0;
      for (uint32_t w = 0; w < 16u; w++) alive_slots += s_live[w];
      if (total) atomicAdd(&totals[0], (unsigned long long)total);
      if (alive_slots) atomicAdd(&totals[1], (unsigned long long)alive_slots);
    }
  }
  __syncthreads();
  uint32_t at = s_base + s_wave[wave] + incl - n;
#pragma unroll
  for (uint32_t k = 0; k < kPer; k++) {
    const uint32_t slot = first + k * 1024u + threadIdx.x;
    uint32_t bits = m[k];
    while (bits) {
      const uint32_t q = (uint32_t)__ffs((int)bits) - 1u;
      bits &= bits - 1u;
      ray_id[at++] = q * nlanes + slot;
    }
  }
  (void)nslots;
  // The slots that are still alive, in ascending order within this block's run: the next generation's logic kernels work from this
  // list - lane i takes slot alive_list[i] - so that in the long tail of a launch (the unit queue is drained, the population thins out
  // over the ~25 generations the longest units still need) their waves are full of live paths instead of mostly dead slots.  While
  // every slot is alive the list is the identity.  (Thread t scans eight CONSECUTIVE slots here: ascending order keeps the logic
  // kernels' plane accesses coalesced.)
  __syncthreads();
  const uint32_t base8 = first + threadIdx.x * kPer;
  uint32_t amask = 0u;
#pragma unroll
  for (uint32_t k = 0; k < kPer; k++) {
    const uint32_t slot = base8 + k;
    if (slot < nlanes && (emit[slot] & 0x80000000u) != 0u) amask |= 1u << k;
  }
  const uint32_t na = (uint32_t)__popc(amask);
  uint32_t ia = na;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)ia, off); if ((int)lane >= off) ia += v; }
  if (lane == 63u) s_wave[wave] = ia;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (uint32_t w = 0; w < 16u; w++) { const uint32_t c = s_wave[w]; s_wave[w] = total; total += c; }
    s_base = total ? atomicAdd(&sc->alive_n[(gen + 1u) & 1u], total) : 0u;
  }
  __syncthreads();
  uint32_t aat = s_base + s_wave[wave] + ia - na;
  while (amask) {
    const uint32_t k = (uint32_t)__ffs((int)amask) - 1u;
    amask &= amask - 1u;
    alive_list[aat++] = base8 + k;
  }
}

// After the last generation: the logic blocks' ray counts into the context's totals {rays, rays elided}.  The number of
// generations the host enqueued is an upper bound derived from the longest chain of batches a unit can need; should it ever be
// too small (a batch type the bound does not know), units are unfinished and their samples stale: `fault` (host-visible, sticky)
// is raised and the next synchronising call of the C ABI fails instead of handing out a wrong image.
__global__ void pt_stream_finish_kernel(unsigned long long* __restrict__ block_counters, uint32_t nblocks, unsigned long long* __restrict__ totals,
                                        const StreamCounters* __restrict__ sc, uint32_t* __restrict__ fault, const uint32_t* dev_cancel) {
  if (threadIdx.x == 0 && sc->done == 0u && *dev_cancel == 0u) atomicOr(fault, 1u);   // (a cancelled launch is cut short on purpose)
  unsigned long long a = 0, b = 0;
  for (uint32_t i = threadIdx.x; i < nblocks; i += blockDim.x) {
    a += block_counters[2 * (size_t)i]; b += block_counters[2 * (size_t)i + 1];
    block_counters[2 * (size_t)i] = 0; block_counters[2 * (size_t)i + 1] = 0;
  }
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
  if ((threadIdx.x & 63u) == 0u) { if (a) atomicAdd(&totals[0], a); if (b) atomicAdd(&totals[1], b); }
}

// STATS build (diagnostic, SRT_CAST_STATS=1): how the waves spend their loop trips.
enum { CS_OUTER = 0, CS_FETCH, CS_INTERIOR_TRIPS, CS_INTERIOR_LANES, CS_LEAF_TRIPS, CS_LEAF_LANES, CS_LEAF_TRIS, CS_OBJECT_TRIPS, CS_OBJECT_LANES,
       CS_T_FETCH, CS_T_INTERIOR, CS_WALKING_LANES, CS_T_LEAF, CS_T_OBJECT, CS_COUNT };

// One pop: the top frame of a lane in FM_UNWIND - the visit rule for the farther child, or Trace::min of the two children
// (flat_pop) - or, with no frame of the current tree left, the end of that tree: a mesh's tree hands over to the object
// phase (Object::hit is finished there), the top-level tree finishes the ray.
// (WALK: every lane is inside a mesh's tree, which starts at frame 0.)
// The pop is flat_pop (pt_flat.h) written on the packed frame words: word 0 = payload << 2 | near_done << 1 | flag.
template <bool WALK>
SRT_DEV void cast_unwind_step(FlatState& F, const LdsStack& stack) {
  const bool plain = WALK ? (F.sp != 0) : flat_plain_frame(F);
  if (plain) {
    uint32_t w0, w1, w2;
    stack.load3(F.sp - 1, w0, w1, w2);
    SRT_PIN_VGPR(w0);                                    // (see flat_pop: the farther child's reference must survive the inlining)
    const float a = __uint_as_float(w1);
    if ((w0 & 2u) == 0u) {
      // back from the nearer child: student/bvh.inl:216 - also visit the farther one iff ...
      if (a < F.ret.dist || (!F.ret.hit && (w0 & 1u) != 0u)) {
        stack.store3(F.sp - 1, (F.ret.obj << 2) | 2u | (F.ret.hit ? 1u : 0u), __float_as_uint(F.ret.dist), F.ret.tri);
        F.cur = (int32_t)w0 >> 2; F.tx = a; F.ty = __uint_as_float(w2);
        F.mode = FM_NODE;
      } else F.sp--;
    } else {
      // back from the farther child: Trace::min(nearer, farther) - the nearer result wins only when strictly closer
      const bool saved = left_wins((w0 & 1u) != 0u, a, F.ret.hit, F.ret.dist);
      F.ret.hit = saved ? true : F.ret.hit;
      F.ret.dist = saved ? a : F.ret.dist;
      F.ret.obj = saved ? (w0 >> 2) : F.ret.obj;
      F.ret.tri = saved ? w2 : F.ret.tri;
      F.sp--;
    }
  } else if (WALK || F.level) F.mode = FM_OBJECT;
  else { F.res0 = F.ret; F.mode = FM_DONE; }
}
// A lane that has just arrived at a BVH<Object> leaf: its objects are next.
SRT_DEV void cast_enter_leaf_objects(FlatState& F) {
  if (F.mode == FM_NODE && F.cur < 0 && F.level == 0u) {
    const uint32_t packed = (uint32_t)~F.cur;
    F.obj_i = packed >> 3; F.obj_end = F.obj_i + (packed & 7u);
    F.acc = flat_no_hit();
    F.mode = FM_OBJECT;
  }
}

// scene.hit for a queue of rays (see the head comment).  Every trip of the loop the wave runs ONE of three phases for the
// lanes that stand there, and the others wait:
//   walk    one step per lane, no inner loops: an interior record (both child boxes, nearer / farther, push) or one pop
//           of the lane's LDS stack - cheap and by far the most frequent;
//   leaf    the <= 4 triangles of a BVH<Triangle> leaf, loaded together, their twelve IEEE quotients refined together;
//   object  Object::hit: ray -> object space, sphere / one-leaf mesh on the spot, or into / out of a mesh's tree.
// The two expensive phases run when enough lanes have piled up in front of them (or nothing else can run), so they execute
// with those lanes instead of with whoever happens to be there.
#ifndef SRT_CAST_OCC
#define SRT_CAST_OCC 4
#endif
#ifndef SRT_CAST_POPS_UNROLLED
#define SRT_CAST_POPS_UNROLLED 1
#endif
#ifndef SRT_CAST_LEAF_SPREAD
#define SRT_CAST_LEAF_SPREAD 1
#endif
#ifndef SRT_CAST_OCC_WALK
#define SRT_CAST_OCC_WALK 5
#endif
// WALK: the entries are walk requests of the streamed sweeps (P.walk_nr > 0) - no lane ever stands in the top-level tree.
template <bool STATS, bool WALK>
__global__ __launch_bounds__(256, WALK ? SRT_CAST_OCC_WALK : SRT_CAST_OCC) void pt_cast_kernel(DScene S, CastParams P) {
  extern __shared__ uint32_t cast_lds[];
#if SRT_CAST_LEAF_SPREAD
  __shared__ uint8_t cast_pairs[4 * 64];                  // leaf phase: owner lane | place in the leaf << 6 of each (ray, triangle) pair, per wave
#endif
  if (P.sc->done != 0u || __hip_atomic_load(P.dev_cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  const uint32_t nrays = *P.nrays;
  if (nrays == 0u) {
    // nothing to cast.  If no slot is alive either and the unit queue is drained, the launch is finished.
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.sc->alive[P.gen & 1u] == 0u && P.sc->queue_head >= (unsigned long long)P.total_units) P.sc->done = 1u;
    return;
  }
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  const LdsStack stack{(lds_u32*)cast_lds + (size_t)wave * P.lds_frames * 3u * 64u + (uint32_t)lane,
                       P.spill + (size_t)blockIdx.x * blockDim.x + threadIdx.x, (int)P.lds_frames, gridDim.x * blockDim.x};
  // this wave's own run of the list and the start of the common pool (P.own_share / 256 of the list is dealt out)
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6), wave_global = blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave;
  const uint32_t own_n = (uint32_t)(((unsigned long long)nrays * P.own_share) >> 8) / nwaves;
  uint32_t own_next = wave_global * own_n, own_end = own_next + own_n;
  const uint32_t pool_base = nwaves * own_n;
  bool pool_done = false;
  FlatState F;                                            // F.mode == FM_DONE: the lane is idle
  bool have = false;                                      // the lane holds a finished ray whose result is not written yet
  bool exhausted = false;                                 // the queue has nothing left for this wave
  uint32_t my_id = 0;
  V3 wo = v3(0, 0, 0), wd = v3(0, 0, 1);                  // the world ray (restored when the lane leaves a mesh)
  float wb0 = 0.0f, wb1 = 0.0f;
  unsigned long long cs[CS_COUNT];
  if (STATS)
    for (int i = 0; i < CS_COUNT; i++) cs[i] = 0;
#define CAST_STAT(i, v) if (STATS) cs[i] += (v)
  for (;;) {
    const bool at_walk = (F.mode == FM_NODE && F.cur >= 0) || F.mode == FM_UNWIND;
    const bool at_leaf = F.mode == FM_NODE && F.cur < 0;   // (a BVH<Triangle> leaf: the others have become objects)
    const bool at_obj = F.mode == FM_OBJECT;
    const unsigned long long idle = __ballot(F.mode == FM_DONE);
    const uint32_t nidle = (uint32_t)__popcll(idle);
    const uint32_t n_walk = (uint32_t)__popcll(__ballot(at_walk)), n_leaf = (uint32_t)__popcll(__ballot(at_leaf)),
                   n_obj = (uint32_t)__popcll(__ballot(at_obj));
    CAST_STAT(CS_OUTER, 1); CAST_STAT(CS_WALKING_LANES, 64u - nidle);
    unsigned long long t0 = 0;
    if (STATS) t0 = __builtin_readcyclecounter();
    const bool run_leaf = n_leaf >= P.leaf_min || (n_walk == 0u && n_leaf > n_obj);
    // Walk requests: a lane whose tree is finished (FM_OBJECT) is RETIRED in the fetch phase itself - world distance, result,
    // and the lane takes the next request in the same trip - instead of waiting for an object phase twice (tree -> mesh,
    // mesh -> ray) and then for the fetch: lanes in FM_OBJECT count as idle for the fetch threshold.
    const uint32_t nret = WALK ? nidle + n_obj : nidle;
    const bool retire = WALK && n_obj > 0u && (n_obj >= P.object_min || (n_walk == 0u && !run_leaf));
    if (nidle == 64u || (!exhausted && nret >= P.fetch_min) || retire) {
      CAST_STAT(CS_FETCH, 1);
      if (WALK && at_obj) {
        // Object::hit's tail for the mesh (flat_exit): the winner's world distance as Trace::transform recomputes it
        bool hit = F.ret.hit; float dist = F.ret.dist; const uint32_t tri = F.ret.tri;
        if (hit && F.xf) {
          const Object& o = S.objects[F.obj_i];
          Ray ray; ray.o = F.co; ray.d = F.cd; ray.b0 = F.b0; ray.b1 = F.b1;
          const TriHit th = tri_hit(cast_load_tri(S, P, tri), ray);
          const V3 pw = mat_point(o.trans, ray_at(ray, th.t));
          const V3 ow = mat_point(o.trans, ray.o);
          dist = norm(pw - ow);
        }
        Hit acc = flat_no_hit();
        fold(acc, hit, dist, F.obj_i, tri);
        F.res0 = acc;
        F.mode = FM_DONE;
      }
      const unsigned long long idle = __ballot(F.mode == FM_DONE);   // (after the retirements)
      const uint32_t nidle = (uint32_t)__popcll(idle);
      if (have && F.mode == FM_DONE) {                    // results out: every idle lane at once
        const Hit h = F.res0;
        uint2 o;
        o.x = __float_as_uint(h.hit ? h.dist : 0.0f);
        o.y = h.hit ? ((h.obj << P.obj_shift) | h.tri) : 0xFFFFFFFFu;
        if (WALK) o.y = h.hit ? h.tri : 0xFFFFFFFFu;
        // (rays, list entries and hits pass through once: non-temporal, so they do not push the tree out of L2)
        __builtin_nontemporal_store(u32x2_t{o.x, o.y}, reinterpret_cast<u32x2_t*>(P.hits) + my_id);
        have = false;
      }
      if (!exhausted) {                                   // new rays in: one atomic per wave
        uint32_t start = 0;
        // Most of the list is dealt out in advance, a contiguous run per wave that the wave walks through by itself; only
        // the last part is a common pool behind one counter.  (Every fetch an atomic on that one word: ~31 000 of them per
        // generation at the ~90 per microsecond a single address takes was a good part of this kernel's time;
        // the pool is taken P.grab entries at a time, into the wave's own run.)
        if (own_next >= own_end && !pool_done) {
          if (lane == 0) start = atomicAdd(P.head, P.grab);
          start = pool_base + (uint32_t)__shfl((int)start, 0);
          if (start >= nrays) pool_done = true;
          else { own_next = start; own_end = start + P.grab < nrays ? start + P.grab : nrays; }
        }
        uint32_t take = 0;
        if (own_next < own_end) {
          take = own_end - own_next < nidle ? own_end - own_next : nidle;
          start = own_next;
          own_next += take;
        }
        exhausted = pool_done && own_next >= own_end;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
        const uint32_t idx = start + rank;
        if (F.mode == FM_DONE && rank < take && idx < nrays) {
          my_id = __builtin_nontemporal_load(P.ray_id + idx);
          const f32x4 ro = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.ray_o) + 2u * (size_t)my_id);
          const f32x4 rd = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.ray_d) + 2u * (size_t)my_id);
          wo = v3(ro.x, ro.y, ro.z); wd = v3(rd.x, rd.y, rd.z); wb0 = ro.w; wb1 = rd.w;
          if (WALK) {
            // a walk request: the lane starts inside the mesh's tree, as flat_object leaves it there (Tri_Mesh::hit ->
            // BVH<Triangle>::hit with times = dist_bounds / dir.norm()); when the tree is done the object phase finishes
            // Object::hit (world distance) into F.acc, which is the result
            const uint32_t m = my_id / P.nlanes / P.walk_nr;
            const uint32_t slot = m == 0u ? P.lazy_obj[0] : (m == 1u ? P.lazy_obj[1] : (m == 2u ? P.lazy_obj[2] : P.lazy_obj[3]));
            const Object& o = S.objects[slot];
            F.res0 = flat_no_hit(); F.ret = F.res0; F.acc = F.res0;
            F.act1 = false; F.act2 = false; F.r = 0u;
            F.co = wo; F.cd = wd; F.b0 = wb0; F.b1 = wb1;
            F.cinv = v3(1.0f / wd.x, 1.0f / wd.y, 1.0f / wd.z);
            F.inv_ok = finite_f(F.cinv.x) && finite_f(F.cinv.y) && finite_f(F.cinv.z);
            const float dn = norm(wd);
            F.tx = wb0 / dn; F.ty = wb1 / dn;
            F.level = 1; F.rec_base = o.rec_base; F.tri_base = o.tri_base; F.xf = o.has_trans != 0u;
            F.obj_i = slot; F.obj_end = slot + 1u;
            F.sp = 0; F.base_sp = 0; F.cur = 0;
            F.mode = FM_NODE;
          } else {
            flat_begin(F, S, wo, wd, wd, wd, wb0, wb1, true, false, false);
            cast_enter_leaf_objects(F);
          }
          have = true;
        }
      }
      CAST_STAT(CS_T_FETCH, __builtin_readcyclecounter() - t0);
      if (__ballot(F.mode != FM_DONE) == 0ull && exhausted) break;
      continue;
    }
    const bool run_obj = !WALK && !run_leaf && (n_obj >= P.object_min || n_walk == 0u);
    if (!run_leaf && !run_obj) {
      CAST_STAT(CS_INTERIOR_TRIPS, 1); CAST_STAT(CS_INTERIOR_LANES, n_walk);
#if SRT_CAST_POPS_UNROLLED
      if (F.mode == FM_UNWIND) cast_unwind_step<WALK>(F, stack);
      if (F.mode == FM_UNWIND) cast_unwind_step<WALK>(F, stack);                  // (a second pop costs less than another trip)
#else
      if (F.mode == FM_UNWIND) {
#pragma nounroll
        for (uint32_t k = 0; k < P.pops && F.mode == FM_UNWIND; k++) cast_unwind_step<WALK>(F, stack);   // (a second pop costs less than another trip)
      }
#endif
      // (pops first, then the interior step for every lane that stands at an interior node NOW - also the ones a pop has just
      //  sent into a farther child: they would otherwise idle through this trip's interior code and come back for the next)
      if (!WALK) cast_enter_leaf_objects(F);
      if (F.mode == FM_NODE && F.cur >= 0) flat_interior<LdsStack, WALK ? 1 : -1>(F, stack, S);
      if (!WALK) cast_enter_leaf_objects(F);
      CAST_STAT(CS_T_INTERIOR, __builtin_readcyclecounter() - t0);
    } else if (run_obj) {
      CAST_STAT(CS_OBJECT_TRIPS, 1); CAST_STAT(CS_OBJECT_LANES, n_obj);
      if (at_obj) {
        if (WALK || F.level) flat_exit(F, S, wo, wd, wd, wd, wb0, wb1);   // back from a mesh's tree: finish its Object::hit
        else if (F.obj_i < F.obj_end) flat_object(F, S);          // the next object of the leaf / list
        if (F.mode == FM_OBJECT && F.obj_i >= F.obj_end) { F.ret = F.acc; F.mode = FM_UNWIND; }
      }
      CAST_STAT(CS_T_OBJECT, __builtin_readcyclecounter() - t0);
    } else {
      if (STATS) {
        unsigned long long t = at_leaf ? (((uint32_t)~F.cur) & 7u) : 0u;
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
        cs[CS_LEAF_TRIPS]++; cs[CS_LEAF_LANES] += n_leaf; cs[CS_LEAF_TRIS] += __shfl(t, 0);
      }
#if SRT_CAST_LEAF_SPREAD
      {
        // BVH<Triangle> leaves, ONE (ray, triangle) test per lane.  A lane at a leaf holds 1..4 triangles, and only ~13 lanes of
        // the 64 stand at a leaf when the phase runs: tested by their own lanes those were four tests at a fifth of the wave.
        // Here the pairs are dealt out over the whole wave - lane w takes pair w: the ray of its owner (nine ds_bpermute) and
        // one triangle - and every owner then folds its own pairs' verdicts in the leaf's order (Trace::min is taken pair by
        // pair either way: same tests, same operands, same order of folding).  At most 64 pairs per phase: the lanes whose pairs
        // do not fit stay at their leaf for the next trip.  (A leaf of 0 or more than 4 triangles - not built by
        // BVH::build with max_leaf_size 4, but legal - is folded by its own lane as before.)
        const uint32_t packed = (uint32_t)~F.cur;
        const uint32_t n = at_leaf ? (packed & 7u) : 0u;
        const bool small = at_leaf && n >= 1u && n <= 4u;
        const uint32_t nn = small ? n : 0u;
        const unsigned long long b0 = __ballot((nn & 1u) != 0u), b1 = __ballot((nn & 2u) != 0u), b2 = __ballot((nn & 4u) != 0u);
        const uint32_t off = __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)) +
                             2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u)) +
                             4u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, 0u));
        const bool sel = small && off + nn <= 64u;             // (a prefix of the leaf lanes: `off` counts every earlier one)
        const unsigned long long selm = __ballot(sel);
        if (selm != 0ull) {
          const int last = 63 - __builtin_clzll(selm);
          const uint32_t npairs = (uint32_t)__shfl((int)(off + nn), last);
          // who owns pair w: the owners write their lane (and the pair's place in the leaf) into the wave's 64 bytes of LDS
          lds_u8* own = (lds_u8*)cast_pairs + wave * 64;
          if (sel) {
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++)
              if (k < nn) own[off + k] = (uint8_t)((uint32_t)lane | (k << 6));
          }
          __builtin_amdgcn_wave_barrier();
          const uint32_t ob = own[lane];
          __builtin_amdgcn_wave_barrier();
          const int owner = (int)(ob & 63u);
          const uint32_t first = F.tri_base + (packed >> 3);
          Ray ray;
          ray.o = v3(__shfl(F.co.x, owner), __shfl(F.co.y, owner), __shfl(F.co.z, owner));
          ray.d = v3(__shfl(F.cd.x, owner), __shfl(F.cd.y, owner), __shfl(F.cd.z, owner));
          ray.b0 = __shfl(F.b0, owner); ray.b1 = __shfl(F.b1, owner);
          const uint32_t tri = (uint32_t)__shfl((int)first, owner) + (ob >> 6);
          TriHit th; th.hit = false; th.dist = 0.0f;
          if ((uint32_t)lane < npairs) {
            const Tri g = cast_load_tri(S, P, tri);
            tri_hit_leafN<1>(&g, ray, &th);
          }
          const int hit_i = th.hit ? 1 : 0;
          Hit acc = flat_no_hit();
#pragma unroll
          for (uint32_t k = 0; k < 4u; k++) {
            const int src = (int)((off + k) & 63u);
            const int hk = __shfl(hit_i, src);
            const float dk = __shfl(th.dist, src);
            if (k < nn) fold(acc, hk != 0, dk, 0, first + k);
          }
          if (sel) { F.ret = acc; F.mode = FM_UNWIND; }
        }
        if (at_leaf && !small) {
          const uint32_t first = F.tri_base + (packed >> 3);
          Ray ray; ray.o = F.co; ray.d = F.cd; ray.b0 = F.b0; ray.b1 = F.b1;
          F.ret = flat_no_hit();
          for (uint32_t i = 0; i < n; i++) {
            const TriHit th = tri_hit(cast_load_tri(S, P, first + i), ray);
            fold(F.ret, th.hit, th.dist, 0, first + i);
          }
          F.mode = FM_UNWIND;
        }
      }
#else
      if (at_leaf) {
        // BVH<Triangle> leaf: fold its triangles in order (spare slots repeat the last one: sane operands, never folded)
        const uint32_t packed = (uint32_t)~F.cur;
        const uint32_t first = F.tri_base + (packed >> 3), n = packed & 7u;
        Ray ray; ray.o = F.co; ray.d = F.cd; ray.b0 = F.b0; ray.b1 = F.b1;
        F.ret = flat_no_hit();
        if (n != 0u && n <= 4u) {
          Tri g[4];
          // (48-byte records: the byte offset by shift-and-add - first * 48 < 2^32 for < 2^26 triangles - instead of a 64-bit multiply-add)
          const char* tb = reinterpret_cast<const char*>(S.tris) + ((((first << 1) + first)) << 4);
#pragma unroll
          for (uint32_t i = 0; i < 4u; i++) g[i] = *reinterpret_cast<const Tri*>(tb + (i < n ? i : n - 1u) * 48u);
          TriHit th[4];
          tri_hit_leaf4(g, ray, th);
#pragma unroll
          for (uint32_t i = 0; i < 4u; i++)
            if (i < n) fold(F.ret, th[i].hit, th[i].dist, 0, first + i);
        } else {
          for (uint32_t i = 0; i < n; i++) {
            const TriHit th = tri_hit(S.tris[first + i], ray);
            fold(F.ret, th.hit, th.dist, 0, first + i);
          }
        }
        F.mode = FM_UNWIND;
      }
#endif
      CAST_STAT(CS_T_LEAF, __builtin_readcyclecounter() - t0);
    }
  }
#undef CAST_STAT
  if (STATS && lane == 0)
    for (int i = 0; i < CS_COUNT; i++) atomicAdd(&P.stats[i], cs[i]);
}

}  // namespace srt

#endif
