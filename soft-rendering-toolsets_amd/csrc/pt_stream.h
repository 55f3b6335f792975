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
  unsigned long long units_done;   // units whose last sample has been stored (or that lie outside the image)
  uint32_t nrays[2];               // rays appended by logic generation g: [g & 1]
  uint32_t cast_head[2];           // next ray of generation g to hand to a cast wave: [g & 1]
};

// Words of a path slot's saved state (planes of `nlanes` words).  The DL build appends its shadow-phase state.
enum {
  SW_FLAGS = 0, SW_PX, SW_PY, SW_SAMPLES, SW_PIXEL_SLOT, SW_PEND0, SW_PEND1, SW_RNG_LO, SW_RNG_HI, SW_ORG, SW_DC = SW_ORG + 3,
  SW_CB0 = SW_DC + 3, SW_CB1, SW_ATT, SW_PDF4 = SW_ATT + 3, SW_PDF_AREA, SW_BASE_WORDS,
  // DL only
  SW_D0 = SW_BASE_WORDS, SW_D1 = SW_D0 + 3, SW_LIGHT_I = SW_D1 + 3, SW_HELD, SW_PL, SW_DA = SW_PL + 3, SW_D6 = SW_DA + 3,
  SW_DCK = SW_D6 + 3, SW_SB1 = SW_DCK + 3, SW_DL_WORDS = SW_SB1 + 3
};

// 12-byte frames in LDS.  Word 0 holds the farther child's reference and "hitboth" until the nearer child has returned, then
// the nearer child's object slot and hit flag (the reference is dead by then): payload << 2 | near_done << 1 | flag.
// References: interior rank < 2^29, leaf ~(first << 3 | count) > -2^29 (checked on the host).
struct LdsStack {
  uint32_t* w;   // this lane's column: word k of frame i at w[(i * 3 + k) * 64]
  SRT_DEV FlatFrame load(int i) const {
    const uint32_t w0 = w[(i * 3 + 0) * 64];
    FlatFrame f;
    f.a = __uint_as_float(w[(i * 3 + 1) * 64]);
    f.b = w[(i * 3 + 2) * 64];
    const bool near_done = (w0 & 2u) != 0;
    f.second = near_done ? 0 : ((int32_t)w0 >> 2);
    f.fl = near_done ? (2u | ((w0 & 1u) << 2) | ((w0 >> 2) << 3)) : (w0 & 1u);
    return f;
  }
  SRT_DEV void store(int i, const FlatFrame& f) const {
    const bool near_done = (f.fl & 2u) != 0;
    w[(i * 3 + 0) * 64] = near_done ? (((f.fl >> 3) << 2) | 2u | ((f.fl >> 2) & 1u)) : (((uint32_t)f.second << 2) | (f.fl & 1u));
    w[(i * 3 + 1) * 64] = __float_as_uint(f.a);
    w[(i * 3 + 2) * 64] = f.b;
  }
};

struct CastParams {
  const float4* ray_o;       // [ray] origin, dist_bounds.x
  const float4* ray_d;       // [ray] direction, dist_bounds.y
  const uint32_t* ray_id;    // [ray] path slot * 4 + batch slot
  uint2* hits;               // [batch slot * nlanes + path slot] {distance bits, object << obj_shift | triangle, or 0xFFFFFFFF}
  uint32_t nlanes;
  const uint32_t* nrays;     // rays of this generation
  uint32_t* head;            // queue head of this generation
  uint32_t depth;            // frames per lane
  uint32_t fetch_min;        // idle lanes of a wave that trigger a fetch
  uint32_t interior_min;     // lanes at interior records that keep the wave in the interior-step loop
  uint32_t obj_shift;
};

// scene.hit for a queue of rays (see the head comment).
__global__ __launch_bounds__(256) void pt_cast_kernel(DScene S, CastParams P) {
  extern __shared__ uint32_t cast_lds[];
  const uint32_t nrays = *P.nrays;
  if (nrays == 0u) return;
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  const LdsStack stack{cast_lds + (size_t)wave * P.depth * 3u * 64u + (uint32_t)lane};
  FlatState F;                                            // F.mode == FM_DONE: the lane is idle
  bool have = false;                                      // the lane holds a finished ray whose result is not written yet
  bool exhausted = false;                                 // the queue has nothing left for this wave
  uint32_t my_id = 0;
  V3 wo = v3(0, 0, 0), wd = v3(0, 0, 1);                  // the world ray (restored when the lane leaves a mesh)
  float wb0 = 0.0f, wb1 = 0.0f;
  for (;;) {
    const unsigned long long idle = __ballot(F.mode == FM_DONE);
    const uint32_t nidle = (uint32_t)__popcll(idle);
    if (nidle >= P.fetch_min || nidle == 64u) {
      if (have && F.mode == FM_DONE) {                    // results out: every idle lane at once
        const Hit h = F.res0;
        uint2 o;
        o.x = __float_as_uint(h.hit ? h.dist : 0.0f);
        o.y = h.hit ? ((h.obj << P.obj_shift) | h.tri) : 0xFFFFFFFFu;
        P.hits[(size_t)(my_id & 3u) * P.nlanes + (my_id >> 2)] = o;
        have = false;
      }
      if (!exhausted) {                                   // new rays in: one atomic per wave
        uint32_t start = 0;
        if (lane == 0) start = atomicAdd(P.head, nidle);
        start = (uint32_t)__shfl((int)start, 0);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
        const uint32_t idx = start + rank;
        if (F.mode == FM_DONE && idx < nrays) {
          const float4 ro = P.ray_o[idx], rd = P.ray_d[idx];
          my_id = P.ray_id[idx];
          wo = v3(ro.x, ro.y, ro.z); wd = v3(rd.x, rd.y, rd.z); wb0 = ro.w; wb1 = rd.w;
          flat_begin(F, S, wo, wd, wd, wd, wb0, wb1, true, false, false);
          have = true;
        }
        if (start + nidle >= nrays) exhausted = true;
      }
      if (__ballot(F.mode != FM_DONE) == 0ull) {
        if (exhausted) break;
        continue;
      }
    }
    // interior records: the bulk of the work and uniform in cost - repeated while enough lanes take part
    for (;;) {
      const bool can = F.mode == FM_NODE && F.cur >= 0;
      if (__ballot(can) == 0ull) break;
      if (can) {
        flat_interior(F, stack, S);
        while (F.mode == FM_UNWIND && flat_plain_frame(F)) flat_pop(F, stack);   // a double miss: straight back to a node
      }
      if ((uint32_t)__popcll(__ballot(F.mode == FM_NODE && F.cur >= 0)) < P.interior_min) break;
    }
    if (F.mode == FM_NODE && F.cur < 0) flat_leaf(F, S);
    if (F.mode == FM_OBJECT) flat_object(F, S);
    while (F.mode == FM_UNWIND) {
      if (flat_plain_frame(F)) flat_pop(F, stack);
      else flat_exit(F, S, wo, wd, wd, wd, wb0, wb1);
    }
  }
}

}  // namespace srt

#endif
