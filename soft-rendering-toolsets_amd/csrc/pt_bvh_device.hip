// BVH<Primitive>::build (student/bvh.inl:35-163) on the device: node arrays and primitive order identical, bit for bit, to the
// host build of pt_scene.cpp (and with it to the reference's), for the large meshes of BASELINE configs[4].
//
// What has to be reproduced is not only the tree: the reference evaluates every candidate plane (three axes x up to nine
// planes) by running std::partition on the node's slice of the primitive array IN PLACE, one after the other, and the order
// in which primitives end up inside the leaves - and with it which of two equally distant triangles wins Trace::min - is the
// product of that whole sequence of permutations.  libstdc++'s partition (bidirectional iterators) swaps the k-th misplaced
// element from the left with the k-th misplaced element from the right, so one partition is: count the primitives in front of
// the plane (mid), rank the "false" ones before mid in ascending order and the "true" ones from mid on in descending order,
// swap equal ranks.  Bounding boxes are folded with std::min / std::max IN INDEX ORDER (the sign of a zero bound depends on
// it): consecutive blocks of the slice are reduced lane by lane, wave by wave, then block by block, earlier operand left.
//
// One workgroup per node of the current level (the reference's node numbering is level order: children are appended as their
// parent is processed), levels one launch after the other; a single block walks a big node's slice in coalesced steps.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <vector>

#include "pt_scene.h"
#include "srt_common.h"

namespace srt {
namespace {

struct DBox { float mn[3], mx[3]; };
struct DNode { float mn[3], mx[3]; uint32_t start, size, l, r; };
static_assert(sizeof(DNode) == sizeof(HostNode), "node layout");

__device__ __forceinline__ float std_minf(float a, float b) { return (b < a) ? b : a; }   // std::min(a, b)
__device__ __forceinline__ float std_maxf(float a, float b) { return (a < b) ? b : a; }   // std::max(a, b)
__device__ __forceinline__ void box_init(DBox& b) { for (int i = 0; i < 3; i++) { b.mn[i] = FLT_MAX; b.mx[i] = -FLT_MAX; } }
// acc.enclose(x) with acc the EARLIER operand
__device__ __forceinline__ void box_enclose(DBox& acc, const DBox& x) {
  for (int i = 0; i < 3; i++) { acc.mn[i] = std_minf(acc.mn[i], x.mn[i]); acc.mx[i] = std_maxf(acc.mx[i], x.mx[i]); }
}
__device__ __forceinline__ float box_area(const DBox& b) {   // lib/bbox.h:50-54
  if (b.mn[0] > b.mx[0] || b.mn[1] > b.mx[1] || b.mn[2] > b.mx[2]) return 0.0f;
  const float ex = b.mx[0] - b.mn[0], ey = b.mx[1] - b.mn[1], ez = b.mx[2] - b.mn[2];
  return 2.0f * (ex * ez + ex * ey + ey * ez);
}
__device__ __forceinline__ float box_center(const DBox& b, int axis) { return (b.mn[axis] + b.mx[axis]) * 0.5f; }

// ordered fold of one box per thread over the block (thread order), result valid in thread 0; `s` holds one box per wave
__device__ void block_fold_box(DBox& v, DBox* s, int lane, int wave, int nwaves) {
  for (int off = 1; off < 64; off <<= 1) {
    DBox o;
    for (int i = 0; i < 3; i++) { o.mn[i] = __shfl_down(v.mn[i], off); o.mx[i] = __shfl_down(v.mx[i], off); }
    if (lane + off < 64) box_enclose(v, o);   // (own = earlier operand)
  }
  // (a shuffle tree in which lane l folds [l, l + 2 off) keeps "earlier left": after the last step lane 0 holds lanes 0..63)
  if (lane == 0) s[wave] = v;
  __syncthreads();
  if (wave == 0 && lane == 0) { DBox a = s[0]; for (int w = 1; w < nwaves; w++) box_enclose(a, s[w]); s[0] = a; }
  __syncthreads();
  v = s[0];
  __syncthreads();
}

__device__ uint32_t block_sum(uint32_t v, uint32_t* s, int lane, int wave, int nwaves) {
  for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_down((int)v, off);
  if (lane == 0) s[wave] = v;
  __syncthreads();
  uint32_t t = 0;
  for (int w = 0; w < nwaves; w++) t += s[w];
  __syncthreads();
  return t;
}

// One std::partition of prim[start, end) by center(axis) < line; returns the number in front (mid - start).
__device__ uint32_t partition_slice(uint32_t* __restrict__ prim, const DBox* __restrict__ boxes, uint32_t start, uint32_t end, int axis, float line,
                                    uint32_t* __restrict__ lpos, uint32_t* __restrict__ rpos, uint32_t* s_u, int lane, int wave, int nwaves) {
  const uint32_t T = blockDim.x;
  uint32_t cnt = 0;
  for (uint32_t i = start + threadIdx.x; i < end; i += T) cnt += box_center(boxes[prim[i]], axis) < line ? 1u : 0u;
  const uint32_t nt = block_sum(cnt, s_u, lane, wave, nwaves);
  const uint32_t mid = start + nt;
  // ranks: misplaced on the left (false before mid) and on the right (true from mid on), both in ascending order
  uint32_t base_l = 0, base_r = 0;
  for (uint32_t b = start; b < end; b += T) {
    const uint32_t i = b + threadIdx.x;
    bool ml = false, mr = false;
    if (i < end) {
      const bool p = box_center(boxes[prim[i]], axis) < line;
      ml = i < mid && !p;
      mr = i >= mid && p;
    }
    const unsigned long long bl = __ballot(ml), br = __ballot(mr);
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (lane == 0) { s_u[wave] = (uint32_t)__popcll(bl); s_u[16 + wave] = (uint32_t)__popcll(br); }
    __syncthreads();
    uint32_t wl = 0, wr = 0, tl = 0, tr = 0;
    for (int w = 0; w < nwaves; w++) { const uint32_t a = s_u[w], c = s_u[16 + w]; if (w < wave) { wl += a; wr += c; } tl += a; tr += c; }
    if (ml) lpos[start + base_l + wl + (uint32_t)__popcll(bl & lt)] = i;
    if (mr) rpos[start + base_r + wr + (uint32_t)__popcll(br & lt)] = i;
    base_l += tl; base_r += tr;
    __syncthreads();
  }
  // swap the k-th from the left with the k-th from the right END (base_l == base_r)
  const uint32_t m = base_l;
  __threadfence_block();
  for (uint32_t k = threadIdx.x; k < m; k += T) {
    const uint32_t a = lpos[start + k], c = rpos[start + (m - 1u - k)];
    const uint32_t pa = prim[a], pc = prim[c];
    prim[a] = pc; prim[c] = pa;
  }
  __syncthreads();
  return nt;
}

struct SplitRec { DBox left, right; uint32_t nl, nr; float line; };

// level pass 1: which nodes of [first, first + count) split, and where their children go (children are appended in node order)
__global__ void bvh_level_plan(const DNode* __restrict__ nodes, uint32_t first, uint32_t count, uint32_t max_leaf, uint32_t next_first,
                               uint32_t* __restrict__ child_at, uint32_t* __restrict__ nsplit) {
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_run;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  if (threadIdx.x == 0) s_run = 0;
  __syncthreads();
  for (uint32_t b = 0; b < count; b += blockDim.x) {
    const uint32_t i = b + threadIdx.x;
    const bool sp = i < count && nodes[first + i].size > max_leaf;
    const unsigned long long m = __ballot(sp);
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int w = 0; w < nwaves; w++) { const uint32_t c = s_w[w]; if (w < wave) before += c; all += c; }
    if (i < count) child_at[i] = sp ? next_first + 2u * (s_run + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) : 0xFFFFFFFFu;
    __syncthreads();
    if (threadIdx.x == 0) s_run += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) *nsplit = s_run;
}

// level pass 2: one block per node
__global__ void bvh_level_split(DNode* __restrict__ nodes, uint32_t first, const uint32_t* __restrict__ child_at, uint32_t* __restrict__ prim,
                                const DBox* __restrict__ boxes, uint32_t* __restrict__ lpos, uint32_t* __restrict__ rpos) {
  const uint32_t node = first + blockIdx.x;
  const uint32_t at = child_at[blockIdx.x];
  if (at == 0xFFFFFFFFu) return;
  __shared__ uint32_t s_u[32];
  __shared__ DBox s_box[16];
  __shared__ SplitRec s_best[3];
  __shared__ float s_cost[3];
  __shared__ float s_plane;
  __shared__ int s_go;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const DNode nd = nodes[node];
  DBox nbox;
  for (int i = 0; i < 3; i++) { nbox.mn[i] = nd.mn[i]; nbox.mx[i] = nd.mx[i]; }
  const uint32_t start = nd.start, end = nd.start + nd.size;
  const uint32_t T = blockDim.x;
  for (int axis = 0; axis < 3; axis++) {
    const float interval = (nbox.mx[axis] - nbox.mn[axis]) / (float)10;
    if (threadIdx.x == 0) {
      s_cost[axis] = FLT_MAX;
      SplitRec e; box_init(e.left); box_init(e.right); e.nl = 0; e.nr = 0; e.line = 0.0f;
      s_best[axis] = e;
      s_plane = nbox.mn[axis] + interval;
      s_go = s_plane < nbox.mx[axis] ? 1 : 0;
    }
    __syncthreads();
    while (s_go) {
      const float plane = s_plane;
      __syncthreads();
      const uint32_t nl = partition_slice(prim, boxes, start, end, axis, plane, lpos, rpos, s_u, lane, wave, nwaves);
      const uint32_t mid = start + nl;
      // left / right boxes: ordered folds over [start, mid) and [mid, end)
      DBox L, R;
      box_init(L); box_init(R);
      for (uint32_t b = start; b < end; b += T) {
        const uint32_t i = b + threadIdx.x;
        DBox l, r;
        box_init(l); box_init(r);
        if (i < end) { const DBox x = boxes[prim[i]]; if (i >= mid) r = x; else l = x; }
        // (an empty operand is the identity of enclose: min(a, FLT_MAX) = a, max(a, -FLT_MAX) = a for finite a)
        block_fold_box(l, s_box, lane, wave, nwaves);
        block_fold_box(r, s_box, lane, wave, nwaves);
        box_enclose(L, l); box_enclose(R, r);
      }
      if (threadIdx.x == 0) {
        const float cost = box_area(L) / box_area(nbox) * (float)(int)nl + box_area(R) / box_area(nbox) * (float)(int)(nd.size - nl) + 1.0f;
        if (cost < s_cost[axis]) {
          s_cost[axis] = cost;
          SplitRec e; e.left = L; e.right = R; e.nl = nl; e.nr = nd.size - nl; e.line = plane;
          s_best[axis] = e;
        }
        s_plane = plane + interval;
        s_go = s_plane < nbox.mx[axis] ? 1 : 0;
      }
      __syncthreads();
    }
    __syncthreads();
  }
  const float lowest = std_minf(s_cost[0], std_minf(s_cost[1], s_cost[2]));
  const int axis = (lowest == s_cost[0]) ? 0 : ((lowest == s_cost[1]) ? 1 : 2);
  const SplitRec best = s_best[axis];
  __syncthreads();
  partition_slice(prim, boxes, start, end, axis, best.line, lpos, rpos, s_u, lane, wave, nwaves);
  if (threadIdx.x == 0) {
    DNode a, b;
    for (int i = 0; i < 3; i++) { a.mn[i] = best.left.mn[i]; a.mx[i] = best.left.mx[i]; b.mn[i] = best.right.mn[i]; b.mx[i] = best.right.mx[i]; }
    a.start = nd.start; a.size = best.nl; a.l = 0; a.r = 0;
    b.start = nd.start + best.nl; b.size = best.nr; b.l = 0; b.r = 0;
    nodes[at] = a; nodes[at + 1] = b;
    nodes[node].l = at; nodes[node].r = at + 1;
  }
}

// the root: Box all; for (b : boxes) all.enclose(b)  (index order), prim = identity
__global__ void bvh_root(DNode* __restrict__ nodes, uint32_t* __restrict__ prim, const DBox* __restrict__ boxes, uint32_t n) {
  __shared__ DBox s_box[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  DBox all;
  box_init(all);
  for (uint32_t b = 0; b < n; b += blockDim.x) {
    const uint32_t i = b + threadIdx.x;
    DBox x;
    box_init(x);
    if (i < n) { x = boxes[i]; prim[i] = i; }
    block_fold_box(x, s_box, lane, wave, nwaves);
    box_enclose(all, x);
  }
  if (threadIdx.x == 0) {
    DNode r;
    for (int i = 0; i < 3; i++) { r.mn[i] = all.mn[i]; r.mx[i] = all.mx[i]; }
    r.start = 0; r.size = n; r.l = 0; r.r = 0;
    nodes[0] = r;
  }
}

#define BVH_HIP(x) do { if ((x) != hipSuccess) { ok = false; goto done; } } while (0)

}  // namespace

// The device builder behind build_bvh (pt_scene.cpp) for large primitive sets.  `boxes6`: n x {mn[3], mx[3]} on the host.
// Returns false when the build does not terminate (as the host build does) or a HIP call fails (out->nodes is then empty and
// the caller falls back to reporting the host builder's verdict).
bool build_bvh_device(const float* boxes6, uint32_t n, uint32_t max_leaf, HostBVH* out) {
  bool ok = true;
  const size_t node_limit = 8ull * n + 64;
  const size_t node_cap = node_limit + 2 * (size_t)n + 8;
  DBox* d_boxes = nullptr; DNode* d_nodes = nullptr;
  uint32_t *d_prim = nullptr, *d_l = nullptr, *d_r = nullptr, *d_child = nullptr, *d_ns = nullptr;
  size_t total = 1;
  uint32_t first = 0, count = 1, levels = 0;
  out->nodes.clear(); out->prim.clear();
  BVH_HIP(hipMalloc(&d_boxes, (size_t)n * sizeof(DBox)));
  BVH_HIP(hipMalloc(&d_nodes, node_cap * sizeof(DNode)));
  BVH_HIP(hipMalloc(&d_prim, (size_t)n * 4)); BVH_HIP(hipMalloc(&d_l, (size_t)n * 4)); BVH_HIP(hipMalloc(&d_r, (size_t)n * 4));
  BVH_HIP(hipMalloc(&d_child, node_cap * 4)); BVH_HIP(hipMalloc(&d_ns, 4));
  BVH_HIP(hipMemcpy(d_boxes, boxes6, (size_t)n * sizeof(DBox), hipMemcpyHostToDevice));
  bvh_root<<<dim3(1), dim3(1024)>>>(d_nodes, d_prim, d_boxes, n);
  while (count) {
    // One launch + one blocking read-back per level.  A usable tree is at most 48 levels deep (kMaxBlasDepth; deeper ones are
    // refused at commit), while a degenerate input - every centre equal - peels one primitive per level: leave those to the
    // host builder's verdict instead of ~4n round trips.
    if (++levels > 512u) { ok = false; goto done; }
    uint32_t nsplit = 0;
    bvh_level_plan<<<dim3(1), dim3(1024)>>>(d_nodes, first, count, max_leaf, (uint32_t)total, d_child, d_ns);
    // block size by the level's width (a proxy for its node sizes): a slice of k primitives is walked k / block steps per pass
    const uint32_t threads = (first == 0 || count <= 64u) ? 1024u : (count <= 4096u ? 256u : 64u);
    bvh_level_split<<<dim3(count), dim3(threads)>>>(d_nodes, first, d_child, d_prim, d_boxes, d_l, d_r);
    BVH_HIP(hipMemcpy(&nsplit, d_ns, 4, hipMemcpyDeviceToHost));
    BVH_HIP(hipGetLastError());
    // the host build gives up when a node is about to split with more than node_limit nodes in the array; the level's last
    // splitting node sees the most (the reference would never return from such a build)
    if (nsplit && total + 2ull * (nsplit - 1u) > node_limit) { ok = false; goto done; }
    first = (uint32_t)total;
    count = 2u * nsplit;
    total += count;
    if (total > node_cap) { ok = false; goto done; }
  }
  out->nodes.resize(total);
  out->prim.resize(n);
  BVH_HIP(hipMemcpy(out->nodes.data(), d_nodes, total * sizeof(DNode), hipMemcpyDeviceToHost));
  BVH_HIP(hipMemcpy(out->prim.data(), d_prim, (size_t)n * 4, hipMemcpyDeviceToHost));
done:
  (void)hipFree(d_boxes); (void)hipFree(d_nodes); (void)hipFree(d_prim); (void)hipFree(d_l); (void)hipFree(d_r); (void)hipFree(d_child); (void)hipFree(d_ns);
  if (!ok) {
    out->nodes.clear(); out->prim.clear();
    (void)hipGetLastError();   // the caller falls back to the host build: a failed hipMalloc here must not surface as the error of a later, unrelated launch
  }
  return ok;
}

}  // namespace srt
