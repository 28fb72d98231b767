// GPU BVH builders, emitted in the reference's BVH layout (include/bvh.h:22-57:
// nodes {first_index, obj_count}, BB_mins_maxes with the sibling boxes of a pair side by side,
// obj_indices, max_depth), so that the same upload, the same kernels and the same oracle walk the trees.
//
// They are NOT the reference's builder: the reference builds a sweep / binned SAH tree on the host
// (src/bvh/sweep_bvh.cpp, bin_bvh.cpp; restated in v-img_amd/host/bvh_build.cpp, which stays the
// default).  They are here for scenes whose geometry changes between frames.  Parity is unaffected by
// construction: GPU kernels and oracle read whatever tree the scene carries.
//
// vimg_hip_build_lbvh: a linear BVH, one primitive per leaf, no cost model (a 1.3x slower walk):
//   1. bounds of the primitive centres                       (atomic min / max on ordered ints)
//   2. 30-bit Morton code of each centre, made unique by appending the primitive index
//   3. radix sort of the 64-bit keys                          (rocPRIM)
//   4. Karras 2012: one thread per internal node finds its key range and its split
//   5. boxes bottom-up: the second thread to arrive at a node merges its children
// vimg_hip_build_ploc: steps 1-3, then
//   4. locally-ordered clustering (Meister & Bittner 2018) bottom-up; a merge also decides, by the
//      reference's cost model, whether its subtree ends as one leaf of up to 8 primitives
//   5. the top of the tree - the 16 K subtrees of largest area - rebuilt top-down by binned SAH
// Both end with
//   6. the reference's layout, a level of the tree per step: breadth-first numbering with siblings
//      adjacent, the sibling-pair box table, obj_indices, the depth.
// Everything runs in kernels; the host launches, reads a counter per round / level, and copies the
// finished arrays out.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/vimg_hip.h"

namespace {

#define LB_TRY(expr)                                 \
  do {                                               \
    hipError_t e_ = (expr);                          \
    if (e_ != hipSuccess) return VIMG_E_DEVICE;      \
  } while (0)

__device__ __forceinline__ uint32_t ordered(float f) {   // order-preserving float -> uint
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float unordered(uint32_t u) {
  const uint32_t v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
  memcpy(&f, &v, 4);
  return f;
}

__global__ void lb_centre_bounds(const float* __restrict__ bounds, uint32_t n, uint32_t* __restrict__ mm) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const float* b = bounds + size_t(i < n ? i : n - 1u) * 6;
  for (int a = 0; a < 3; ++a) {
    uint32_t lo = ordered((b[a] + b[3 + a]) * 0.5f), hi = lo;
    for (int o = 32; o; o >>= 1) lo = min(lo, __shfl_xor(lo, o)), hi = max(hi, __shfl_xor(hi, o));   // one pair of atomics per wave
    if ((threadIdx.x & 63u) == 0u) atomicMin(&mm[a], lo), atomicMax(&mm[3 + a], hi);
  }
}

__device__ __forceinline__ uint32_t expand10(uint32_t v) {   // 10 bits -> every third bit
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__global__ void lb_morton(const float* __restrict__ bounds, uint32_t n, const uint32_t* __restrict__ mm,
                          unsigned long long* __restrict__ keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* b = bounds + size_t(i) * 6;
  uint32_t code = 0;
  for (int a = 0; a < 3; ++a) {
    const float lo = unordered(mm[a]), hi = unordered(mm[3 + a]);
    const float c = (b[a] + b[3 + a]) * 0.5f;
    float t = hi > lo ? (c - lo) / (hi - lo) : 0.f;
    t = fminf(fmaxf(t * 1024.f, 0.f), 1023.f);
    code |= expand10(static_cast<uint32_t>(t)) << (2 - a);
  }
  keys[i] = (static_cast<unsigned long long>(code) << 32) | i;
}

__device__ __forceinline__ int delta(const unsigned long long* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  return __clzll(keys[i] ^ keys[j]);   // keys are unique (index in the low word)
}

// children / parents: a child reference with bit 31 set is a leaf (sorted position in the low
// bits), otherwise an internal node index
__global__ void lb_radix_tree(const unsigned long long* __restrict__ keys, int n, uint32_t* __restrict__ left,
                              uint32_t* __restrict__ right, uint32_t* __restrict__ parent_internal,
                              uint32_t* __restrict__ parent_leaf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    if (t == 1) break;
  }
  const int gamma = i + s * d + (d < 0 ? -1 : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const uint32_t lc = (lo == gamma) ? (0x80000000u | uint32_t(gamma)) : uint32_t(gamma);
  const uint32_t rc = (hi == gamma + 1) ? (0x80000000u | uint32_t(gamma + 1)) : uint32_t(gamma + 1);
  left[i] = lc;
  right[i] = rc;
  if (lc & 0x80000000u) parent_leaf[gamma] = uint32_t(i); else parent_internal[gamma] = uint32_t(i);
  if (rc & 0x80000000u) parent_leaf[gamma + 1] = uint32_t(i); else parent_internal[gamma + 1] = uint32_t(i);
}

__global__ void lb_boxes(const unsigned long long* __restrict__ keys, const float* __restrict__ bounds, int n,
                         const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                         const uint32_t* __restrict__ parent_internal, const uint32_t* __restrict__ parent_leaf,
                         float* __restrict__ leaf_box, float* __restrict__ node_box, uint32_t* __restrict__ visits) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t prim = static_cast<uint32_t>(keys[k] & 0xffffffffull);
  float box[6];
  for (int a = 0; a < 6; ++a) box[a] = leaf_box[size_t(k) * 6 + a] = bounds[size_t(prim) * 6 + a];
  if (n == 1) return;
  uint32_t node = parent_leaf[k];
  for (;;) {
    __threadfence();
    if (atomicAdd(&visits[node], 1u) == 0u) return;   // the sibling subtree is not done yet
    // second arrival: both children are final
    const uint32_t c[2] = {left[node], right[node]};
    for (int s = 0; s < 2; ++s) {
      const float* cb = (c[s] & 0x80000000u) ? leaf_box + size_t(c[s] & 0x7fffffffu) * 6
                                             : node_box + size_t(c[s]) * 6;
      for (int a = 0; a < 3; ++a) {
        const float mn = __hip_atomic_load(cb + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float mx = __hip_atomic_load(cb + 3 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s == 0) { box[a] = mn; box[3 + a] = mx; }
        else { box[a] = fminf(box[a], mn); box[3 + a] = fmaxf(box[3 + a], mx); }
      }
    }
    for (int a = 0; a < 6; ++a)
      __hip_atomic_store(node_box + size_t(node) * 6 + a, box[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (node == 0) return;
    node = parent_internal[node];
  }
}

struct Buf {
  void* p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
  template <typename T> T* as() { return static_cast<T*>(p); }
};

// ---- a binary tree over the sorted leaves -> the reference layout (include/bvh.h:22-57): breadth-first
// numbering with the two children of a node adjacent, the sibling-pair box table, obj_indices with
// every leaf's primitives contiguous, max_depth.  One level of the tree per step: the level's entries in
// order, a scan over them numbers the children (the k-th node with children, in breadth-first order, owns
// 1 + 2k and 2 + 2k) and places the leaves' primitives; the host only reads each level's two totals.
// Tree ids: [0, n) the sorted leaves, n + i the internal node i.  `lbvh_refs`: children are stored the
// radix tree's way (bit 31 = leaf at that sorted position, else internal node index).  `as_leaf[id]`: the
// subtree of `nprims[id]` primitives is emitted as ONE leaf (the builder's decision: ploc_merge).
struct EmitTree {
  const uint32_t* left;
  const uint32_t* right;
  const float* box;
  const uint32_t* nprims;
  const uint32_t* as_leaf;
  const unsigned long long* keys;   // sorted keys: the primitive in the low word
  uint32_t n, lbvh_refs;
};
__device__ __forceinline__ uint32_t emit_child(const EmitTree& t, uint32_t id, bool second) {
  const uint32_t r = (second ? t.right : t.left)[id - t.n];
  if (!t.lbvh_refs) return r;
  return (r & 0x80000000u) ? (r & 0x7fffffffu) : r + t.n;
}
__device__ __forceinline__ bool emit_is_leaf(const EmitTree& t, uint32_t id) { return id < t.n || (t.as_leaf && t.as_leaf[id] != 0u); }
// (nodes with children << 32 | primitives of leaves) per entry of the level
__global__ void emit_classify(EmitTree t, const uint2* __restrict__ level, uint32_t size, unsigned long long* __restrict__ packed) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= size) return;
  const uint32_t id = level[q].x;
  packed[q] = emit_is_leaf(t, id) ? static_cast<unsigned long long>(id < t.n ? 1u : t.nprims[id]) : (1ull << 32);
}
__global__ void __launch_bounds__(256)
emit_write(EmitTree t, const uint2* __restrict__ level, uint32_t size, const unsigned long long* __restrict__ packed,
           const unsigned long long* __restrict__ scanned, uint32_t child_base, uint32_t prim_base, uint2* __restrict__ next_level,
           VimgBVHNode* __restrict__ nodes, float* __restrict__ bb, uint32_t* __restrict__ obj_indices, unsigned long long* __restrict__ totals) {
  __shared__ uint32_t stack_all[256][9];   // a leaf's subtree holds 8 primitives at most
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= size) return;
  const uint32_t id = level[q].x, out = level[q].y;
  const unsigned long long before = scanned[q];
  if (q == size - 1u) *totals = before + packed[q];
  auto put = [&](size_t row, const float* v) { bb[row * 3 + 0] = v[0], bb[row * 3 + 1] = v[1], bb[row * 3 + 2] = v[2]; };
  if (out == 0u) put(0, t.box + size_t(id) * 6), put(2, t.box + size_t(id) * 6 + 3);   // the root
  if (!emit_is_leaf(t, id)) {
    const uint32_t first_child = child_base + 2u * static_cast<uint32_t>(before >> 32);
    const uint32_t rank = (first_child - child_base);
    nodes[out] = VimgBVHNode{first_child, 0u};
    const uint32_t c0 = emit_child(t, id, false), c1 = emit_child(t, id, true);
    const size_t base = size_t(first_child) * 2 + 2;   // {Lmin, Rmin, Lmax, Rmax}
    put(base + 0, t.box + size_t(c0) * 6), put(base + 1, t.box + size_t(c1) * 6);
    put(base + 2, t.box + size_t(c0) * 6 + 3), put(base + 3, t.box + size_t(c1) * 6 + 3);
    next_level[rank] = uint2{c0, first_child}, next_level[rank + 1u] = uint2{c1, first_child + 1u};
    return;
  }
  uint32_t cursor = prim_base + static_cast<uint32_t>(before & 0xffffffffull);
  nodes[out] = VimgBVHNode{cursor, id < t.n ? 1u : t.nprims[id]};
  uint32_t* stack = stack_all[threadIdx.x];
  uint32_t sp = 0;
  stack[sp++] = id;
  while (sp != 0u) {   // depth first, left before right
    const uint32_t r = stack[--sp];
    if (r < t.n) {
      obj_indices[cursor++] = static_cast<uint32_t>(t.keys[r] & 0xffffffffull);
    } else {
      if (sp + 2u > 9u) break;   // (unreachable: such a subtree has 8 leaves at most)
      stack[sp++] = emit_child(t, r, true);
      stack[sp++] = emit_child(t, r, false);
    }
  }
}

int emit_reference_layout(const EmitTree& t, uint32_t root_id, uint32_t* num_nodes, uint32_t* max_depth, VimgBVHNode* nodes, float* bb,
                          uint32_t* obj_indices) {
  const uint32_t n = t.n, threads = 256;
  const size_t max_nodes = size_t(2) * n - 1, bb_rows = 2 * max_nodes + 3;
  Buf d_level[2], d_packed, d_scanned, d_totals, d_nodes, d_bb, d_obj, d_scan;
  for (auto& b : d_level) LB_TRY(hipMalloc(&b.p, size_t(n) * sizeof(uint2)));
  LB_TRY(hipMalloc(&d_packed.p, size_t(n) * 8));
  LB_TRY(hipMalloc(&d_scanned.p, size_t(n) * 8));
  LB_TRY(hipMalloc(&d_totals.p, 8));
  LB_TRY(hipMalloc(&d_nodes.p, max_nodes * sizeof(VimgBVHNode)));
  LB_TRY(hipMalloc(&d_bb.p, bb_rows * 3 * sizeof(float)));
  LB_TRY(hipMalloc(&d_obj.p, size_t(n) * 4));
  size_t scan_bytes = 0;
  LB_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, d_packed.as<unsigned long long>(), d_scanned.as<unsigned long long>(), 0ull, n,
                                 rocprim::plus<unsigned long long>()));
  LB_TRY(hipMalloc(&d_scan.p, std::max<size_t>(scan_bytes, 16)));
  const uint2 first{root_id, 0u};
  LB_TRY(hipMemcpy(d_level[0].p, &first, sizeof(first), hipMemcpyHostToDevice));
  LB_TRY(hipMemsetAsync(d_bb.p, 0, 4 * 3 * sizeof(float), 0));   // (rows 1 and 3 belong to no node)
  uint32_t size = 1, child_base = 1, prim_base = 0, depth = 0;
  int cur = 0;
  while (size != 0u) {
    if (++depth > 4096u) return VIMG_E_DEVICE;
    const uint32_t blocks = (size + threads - 1) / threads;
    hipLaunchKernelGGL(emit_classify, dim3(blocks), dim3(threads), 0, 0, t, d_level[cur].as<uint2>(), size, d_packed.as<unsigned long long>());
    LB_TRY(rocprim::exclusive_scan(d_scan.p, scan_bytes, d_packed.as<unsigned long long>(), d_scanned.as<unsigned long long>(), 0ull, size,
                                   rocprim::plus<unsigned long long>()));
    hipLaunchKernelGGL(emit_write, dim3(blocks), dim3(threads), 0, 0, t, d_level[cur].as<uint2>(), size, d_packed.as<unsigned long long>(),
                       d_scanned.as<unsigned long long>(), child_base, prim_base, d_level[cur ^ 1].as<uint2>(), d_nodes.as<VimgBVHNode>(),
                       d_bb.as<float>(), d_obj.as<uint32_t>(), d_totals.as<unsigned long long>());
    unsigned long long totals = 0;
    LB_TRY(hipMemcpy(&totals, d_totals.p, 8, hipMemcpyDeviceToHost));   // (also the level's synchronisation)
    const uint32_t with_children = static_cast<uint32_t>(totals >> 32);
    child_base += 2u * with_children, prim_base += static_cast<uint32_t>(totals & 0xffffffffull);
    if (child_base > max_nodes || prim_base > n) return VIMG_E_DEVICE;
    size = 2u * with_children;
    cur ^= 1;
  }
  LB_TRY(hipGetLastError());
  if (prim_base != n) return VIMG_E_DEVICE;   // every primitive sits in exactly one leaf
  *num_nodes = child_base;
  *max_depth = depth;
  LB_TRY(hipMemcpy(nodes, d_nodes.p, size_t(child_base) * sizeof(VimgBVHNode), hipMemcpyDeviceToHost));
  LB_TRY(hipMemcpy(bb, d_bb.p, (size_t(2) * child_base + 2) * 3 * sizeof(float), hipMemcpyDeviceToHost));
  LB_TRY(hipMemcpy(obj_indices, d_obj.p, size_t(n) * 4, hipMemcpyDeviceToHost));
  return VIMG_OK;
}

// ---- PLOC (parallel locally-ordered clustering, Meister & Bittner 2018): bottom-up agglomeration
// over the Morton-sorted clusters.  Per round: every cluster looks at its 2 * R neighbours in the
// array for the one whose union with it has the smallest surface area; mutual nearest neighbours
// merge into a new node, everything is compacted (order kept), until one cluster is left.  The
// trees come close to a top-down SAH build where an LBVH, whose splits only look at Morton bits,
// costs the renderer a fifth of its rate.
constexpr int PLOC_R = 12;   // default search radius (tools: VIMG_PLOC_R)
__device__ __forceinline__ float union_half_area(const float* a, const float* b) {
  const float dx = fmaxf(a[3], b[3]) - fminf(a[0], b[0]);
  const float dy = fmaxf(a[4], b[4]) - fminf(a[1], b[1]);
  const float dz = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
  return dx * dy + dx * dz + dy * dz;
}
// node ids: [0, n) the sorted leaves, [n, 2n - 1) the internal nodes in creation order
// per-id records beside the box: primitives below, SAH cost of the subtree (traversal 0.5, intersection 1:
// include/bvh.h:17-20), "this subtree ends as one leaf", parent, first sorted position below (the
// canonical order of disjoint subtrees)
struct PlocRec {
  uint32_t* nprims;
  float* cost;
  uint32_t* as_leaf;
  uint32_t* parent;
  uint32_t* firstpos;
};
__global__ void ploc_leaf_boxes(const unsigned long long* __restrict__ keys, const float* __restrict__ bounds, uint32_t n,
                                float* __restrict__ box, uint32_t* __restrict__ cluster, PlocRec rec) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t prim = static_cast<uint32_t>(keys[k] & 0xffffffffull);
  for (int a = 0; a < 6; ++a) box[size_t(k) * 6 + a] = bounds[size_t(prim) * 6 + a];
  cluster[k] = k;
  rec.nprims[k] = 1u, rec.cost[k] = 1.f, rec.as_leaf[k] = 0u, rec.parent[k] = 0xffffffffu, rec.firstpos[k] = k;
}
__device__ __forceinline__ float half_area_of(const float* b) {
  const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
  return dx * dy + dx * dz + dy * dz;
}
__global__ void ploc_nearest(const uint32_t* __restrict__ cluster, const float* __restrict__ box, uint32_t c,
                             uint32_t radius, uint32_t* __restrict__ nn) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  float mine[6];
  for (int a = 0; a < 6; ++a) mine[a] = box[size_t(cluster[i]) * 6 + a];
  const uint32_t lo = i > radius ? i - radius : 0u, hi = (i + radius < c - 1u) ? i + radius : c - 1u;
  float best = __builtin_huge_valf();
  uint32_t best_j = i;
  for (uint32_t j = lo; j <= hi; ++j) {
    if (j == i) continue;
    const float sa = union_half_area(mine, box + size_t(cluster[j]) * 6);
    if (sa < best) best = sa, best_j = j;   // ties: the lower index (both sides see the same order)
  }
  nn[i] = best_j;
}
__global__ void ploc_merge(const uint32_t* __restrict__ cluster, const uint32_t* __restrict__ nn, uint32_t c, uint32_t n,
                           float* __restrict__ box, uint32_t* __restrict__ left, uint32_t* __restrict__ right,
                           uint32_t* __restrict__ next_node, uint32_t* __restrict__ merged, uint32_t* __restrict__ keep,
                           PlocRec rec) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const uint32_t j = nn[i];
  uint32_t id = cluster[i], k = 1u;
  if (j != i && nn[j] == i) {
    if (i < j) {
      const uint32_t a = cluster[i], b = cluster[j];
      id = n + atomicAdd(next_node, 1u);
      left[id - n] = a;
      right[id - n] = b;
      float u[6];
      for (int x = 0; x < 3; ++x) {
        u[x] = box[size_t(id) * 6 + x] = fminf(box[size_t(a) * 6 + x], box[size_t(b) * 6 + x]);
        u[3 + x] = box[size_t(id) * 6 + 3 + x] = fmaxf(box[size_t(a) * 6 + 3 + x], box[size_t(b) * 6 + 3 + x]);
      }
      // both children are final: the subtree's primitives and cost, and whether it ends as one leaf (up to
      // 8 primitives, where a leaf is no dearer than the split: src/bvh/sweep_bvh.cpp:140-147)
      const float area = half_area_of(u);
      const uint32_t np = rec.nprims[a] + rec.nprims[b];
      float split = 0.5f;
      split += (area > 0.f ? half_area_of(box + size_t(a) * 6) / area : 1.f) * rec.cost[a];
      split += (area > 0.f ? half_area_of(box + size_t(b) * 6) / area : 1.f) * rec.cost[b];
      const float leaf = static_cast<float>(np);
      const bool ends = np <= 8u && leaf <= split;
      rec.nprims[id] = np, rec.cost[id] = ends ? leaf : split, rec.as_leaf[id] = ends ? 1u : 0u;
      rec.parent[id] = 0xffffffffu, rec.parent[a] = id, rec.parent[b] = id;
      rec.firstpos[id] = min(rec.firstpos[a], rec.firstpos[b]);
    } else {
      k = 0u;   // absorbed by its partner
    }
  }
  merged[i] = id;
  keep[i] = k;
}
__global__ void ploc_compact(const uint32_t* __restrict__ merged, const uint32_t* __restrict__ keep,
                             const uint32_t* __restrict__ pos, uint32_t c, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  if (keep[i]) out[pos[i]] = merged[i];
}

// ---- the top of the agglomerated tree, rebuilt top-down ON THE GPU.  Agglomeration decides well near the
// leaves and poorly near the root (its last merges join whatever is left), so the tree is cut at the
// subtrees of largest surface area - a threshold cut: a box contains its children's, so "always open the
// largest" opens exactly the nodes above an area - and a binned surface-area-heuristic tree is built over
// the cut's subtrees (cost of a split = area x primitives on each side, 32 bins on each of the three
// axes; the quantity src/bvh/sweep_bvh.cpp:7-49 sweeps and src/bvh/bin_bvh.cpp bins).  One launch per
// level of the new top, one wave per node of the level; no launch waits for another's data.
constexpr uint32_t kTopItems = 16384;   // subtrees the top is rebuilt over (the cut's size)
constexpr int TOP_BINS = 32;   // (64 bins, or a cut of 64 K subtrees: the same trees within noise)
constexpr uint32_t TOP_SAH_LEVELS = 40;   // deeper levels split at the median (a bound on the launches)
struct __attribute__((aligned(16))) TopItem {
  float lo[3];
  uint32_t id;      // PLOC id of the subtree's root
  float hi[3];
  uint32_t prims;
};
struct TopJob { uint32_t lo, hi, node, pad; };

// keys of the cut: the half area of every internal node that may be opened (more than 8 primitives, so
// that "open" is closed towards the root like the areas are), 0 for the others
__global__ void top_keys(const float* __restrict__ box, const uint32_t* __restrict__ nprims, uint32_t n, uint32_t* __restrict__ key) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i + 1 >= n) return;
  const uint32_t id = n + i;
  key[i] = nprims[id] > 8u ? __float_as_uint(half_area_of(box + size_t(id) * 6)) : 0u;   // (areas are >= 0: the bits order them)
}
// the cut: subtrees that stay closed under an opened parent, at their first sorted position (disjoint
// subtrees have different ones: the order of the cut does not depend on the order nodes were created in)
__global__ void top_mark(const uint32_t* __restrict__ key, const uint32_t* __restrict__ sorted_desc, uint32_t opened, uint32_t n,
                         const uint32_t* __restrict__ parent, const uint32_t* __restrict__ firstpos,
                         uint32_t* __restrict__ flag, uint32_t* __restrict__ at) {
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id + 1 >= 2u * n) return;
  const uint32_t tau = max(sorted_desc[opened - 1u], 1u);
  const uint32_t p = parent[id];
  if (p == 0xffffffffu) return;   // the root
  const bool open_p = key[p - n] >= tau, open_me = id >= n && key[id - n] >= tau;
  if (open_p && !open_me) flag[firstpos[id]] = 1u, at[firstpos[id]] = id;
}
__global__ void top_items(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ at,
                          uint32_t n, const float* __restrict__ box, const uint32_t* __restrict__ nprims, TopItem* __restrict__ items) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n || !flag[k]) return;
  const uint32_t id = at[k];
  TopItem it;
  for (int a = 0; a < 3; ++a) it.lo[a] = box[size_t(id) * 6 + a], it.hi[a] = box[size_t(id) * 6 + 3 + a];
  it.id = id, it.prims = nprims[id];
  items[pos[k]] = it;
}

__device__ __forceinline__ float wave_min_f(float v) {
  for (int o = 32; o; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
  for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ uint32_t wave_sum_u(uint32_t v) {
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One level of the new top: wave j splits the item range of job j into two (written, partitioned, to
// `out`), makes the node, and appends a job for every side of more than one item.
__global__ void __launch_bounds__(256)
top_level(const TopItem* __restrict__ in, TopItem* __restrict__ out, const TopJob* __restrict__ jobs_in, TopJob* __restrict__ jobs_out,
          uint32_t* __restrict__ counts, uint32_t level, uint32_t n, uint32_t* __restrict__ next_node, float* __restrict__ box,
          uint32_t* __restrict__ left, uint32_t* __restrict__ right, uint32_t* __restrict__ nprims, uint32_t* __restrict__ as_leaf) {
  // bins of a wave: [axis][bin]{lo xyz, hi xyz (order-preserving integers), items, primitives}
  __shared__ uint32_t bins_all[4][3][TOP_BINS][8];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t j = blockIdx.x * 4u + wave;
  if (j >= counts[level]) return;   // (wave-uniform; nothing below synchronises across waves)
  uint32_t (*bins)[TOP_BINS][8] = bins_all[wave];
  const TopJob job = jobs_in[j];
  const uint32_t lo = job.lo, hi = job.hi, count = hi - lo;
  // pass 1: the node's box, its primitives, the bounds of the centres
  float bl[3] = {3.4e38f, 3.4e38f, 3.4e38f}, bh[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  float cl[3] = {3.4e38f, 3.4e38f, 3.4e38f}, ch[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  uint32_t np = 0;
  for (uint32_t i = lo + lane; i < hi; i += 64u) {
    const TopItem it = in[i];
    for (int a = 0; a < 3; ++a) {
      const float c = 0.5f * (it.lo[a] + it.hi[a]);
      bl[a] = fminf(bl[a], it.lo[a]), bh[a] = fmaxf(bh[a], it.hi[a]);
      cl[a] = fminf(cl[a], c), ch[a] = fmaxf(ch[a], c);
    }
    np += it.prims;
  }
  for (int a = 0; a < 3; ++a) bl[a] = wave_min_f(bl[a]), bh[a] = wave_max_f(bh[a]), cl[a] = wave_min_f(cl[a]), ch[a] = wave_max_f(ch[a]);
  np = wave_sum_u(np);
  if (lane == 0) {
    for (int a = 0; a < 3; ++a) box[size_t(job.node) * 6 + a] = bl[a], box[size_t(job.node) * 6 + 3 + a] = bh[a];
    nprims[job.node] = np, as_leaf[job.node] = 0u;
  }
  float scale[3];
  for (int a = 0; a < 3; ++a) scale[a] = ch[a] > cl[a] ? float(TOP_BINS) / (ch[a] - cl[a]) : 0.f;
  auto bin_of = [&](const TopItem& it, int a) -> uint32_t {
    const float c = 0.5f * (it.lo[a] + it.hi[a]);
    const int b = static_cast<int>((c - cl[a]) * scale[a]);
    return static_cast<uint32_t>(b < 0 ? 0 : (b > TOP_BINS - 1 ? TOP_BINS - 1 : b));
  };
  // pass 2: bins, then the cheapest of the 3 x 31 splits
  int best_axis = -1;
  uint32_t best_split = 0, n_left = count / 2u;
  if (level < TOP_SAH_LEVELS && count > 2u) {
    for (uint32_t k = lane; k < 3u * TOP_BINS; k += 64u) {
      uint32_t* b = bins[k / TOP_BINS][k % TOP_BINS];
      b[0] = b[1] = b[2] = 0xffffffffu, b[3] = b[4] = b[5] = 0u, b[6] = b[7] = 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (uint32_t i = lo + lane; i < hi; i += 64u) {
      const TopItem it = in[i];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        uint32_t* b = bins[a][bin_of(it, a)];
        for (int x = 0; x < 3; ++x) atomicMin(&b[x], ordered(it.lo[x])), atomicMax(&b[3 + x], ordered(it.hi[x]));
        atomicAdd(&b[6], 1u), atomicAdd(&b[7], it.prims);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    unsigned long long best = ~0ull;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (!(scale[a] > 0.f) || lane >= uint32_t(TOP_BINS - 1)) continue;
      float l[6] = {3.4e38f, 3.4e38f, 3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f}, r[6] = {3.4e38f, 3.4e38f, 3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
      uint32_t li = 0, ri = 0, lp = 0, rp = 0;
      for (uint32_t b = 0; b < uint32_t(TOP_BINS); ++b) {
        const uint32_t* q = bins[a][b];
        if (q[6] == 0u) continue;
        const bool on_left = b <= lane;
        for (int x = 0; x < 3; ++x) {
          const float vlo = unordered(q[x]), vhi = unordered(q[3 + x]);
          l[x] = on_left ? fminf(l[x], vlo) : l[x], l[3 + x] = on_left ? fmaxf(l[3 + x], vhi) : l[3 + x];
          r[x] = on_left ? r[x] : fminf(r[x], vlo), r[3 + x] = on_left ? r[3 + x] : fmaxf(r[3 + x], vhi);
        }
        li += on_left ? q[6] : 0u, lp += on_left ? q[7] : 0u, ri += on_left ? 0u : q[6], rp += on_left ? 0u : q[7];
      }
      if (li == 0u || ri == 0u) continue;
      const float cost = half_area_of(l) * float(lp) + half_area_of(r) * float(rp);
      const unsigned long long key = (static_cast<unsigned long long>(__float_as_uint(cost)) << 32) | (uint32_t(a) * 64u + lane);
      best = key < best ? key : best;
    }
    for (int o = 32; o; o >>= 1) {
      const unsigned long long other = __shfl_xor(best, o);
      best = other < best ? other : best;
    }
    if (best != ~0ull) {
      best_axis = static_cast<int>((best & 0xffffffffull) / 64u), best_split = static_cast<uint32_t>(best & 63ull);
      n_left = 0;
      for (uint32_t b = 0; b <= best_split; ++b) n_left += bins_all[wave][0][0][(uint32_t(best_axis) * TOP_BINS + b) * 8u + 6u];
    }
  }
  // pass 3: the range, partitioned (order kept on each side), into `out`
  uint32_t done_l = 0, done_r = 0, id_l = 0, id_r = 0;
  bool have_l = false, have_r = false;
  auto partition = [&](auto axis_c) {   // (the axis as a constant: no register is indexed)
    constexpr int AX = decltype(axis_c)::value;
    for (uint32_t base = lo; base < hi; base += 64u) {
      const uint32_t i = base + lane;
      const bool valid = i < hi;
      const TopItem it = in[valid ? i : hi - 1u];
      bool goes_left;
      if constexpr (AX >= 0) goes_left = valid && bin_of(it, AX) <= best_split;
      else goes_left = valid && (i - lo) < n_left;
      const unsigned long long ml = __ballot(goes_left), mr = __ballot(valid && !goes_left);
      const unsigned long long below = (1ull << lane) - 1ull;
      if (valid) {
        const uint32_t pos = goes_left ? lo + done_l + uint32_t(__popcll(ml & below)) : lo + n_left + done_r + uint32_t(__popcll(mr & below));
        out[pos] = it;
        if (pos == lo) id_l = it.id, have_l = true;
        if (pos == lo + n_left) id_r = it.id, have_r = true;
      }
      done_l += uint32_t(__popcll(ml)), done_r += uint32_t(__popcll(mr));
    }
  };
  if (best_axis == 0) partition(std::integral_constant<int, 0>{});
  else if (best_axis == 1) partition(std::integral_constant<int, 1>{});
  else if (best_axis == 2) partition(std::integral_constant<int, 2>{});
  else partition(std::integral_constant<int, -1>{});
  // the children: a side of one item is that subtree, a longer one a new node and a job of the next level
  const unsigned long long wl = __ballot(have_l), wr = __ballot(have_r);
  id_l = __shfl(id_l, wl ? __ffsll(static_cast<long long>(wl)) - 1 : 0);
  id_r = __shfl(id_r, wr ? __ffsll(static_cast<long long>(wr)) - 1 : 0);
  if (lane == 0) {
    auto child_of = [&](uint32_t from, uint32_t to, uint32_t only) -> uint32_t {
      if (to - from == 1u) return only;
      const uint32_t id = n + atomicAdd(next_node, 1u);
      jobs_out[atomicAdd(&counts[level + 1u], 1u)] = TopJob{from, to, id, 0u};
      return id;
    };
    const uint32_t child0 = child_of(lo, lo + n_left, id_l), child1 = child_of(lo + n_left, hi, id_r);
    left[job.node - n] = child0, right[job.node - n] = child1;
  }
}

}  // namespace

extern "C" int vimg_hip_build_lbvh(uint32_t n, const float* bounds6, uint32_t* num_nodes,
                                   uint32_t* max_depth, VimgBVHNode* nodes, float* bb,
                                   uint32_t* obj_indices) {
  if (!bounds6 || !num_nodes || !max_depth || !nodes || !bb || !obj_indices || n == 0 || n > (1u << 25))
    return VIMG_E_INVALID;
  if (vimg_hip_device_count() <= 0) return VIMG_E_DEVICE;
  const uint32_t threads = 256, blocks = (n + threads - 1) / threads;
  Buf d_bounds, d_mm, d_keys, d_keys2, d_left, d_right, d_pi, d_pl, d_box, d_visits, d_tmp;
  LB_TRY(hipMalloc(&d_bounds.p, size_t(n) * 6 * sizeof(float)));
  LB_TRY(hipMalloc(&d_mm.p, 6 * sizeof(uint32_t)));
  LB_TRY(hipMalloc(&d_keys.p, size_t(n) * 8));
  LB_TRY(hipMalloc(&d_keys2.p, size_t(n) * 8));
  LB_TRY(hipMalloc(&d_left.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_right.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_pi.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_pl.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_box.p, size_t(2) * n * 6 * sizeof(float)));   // the leaves' boxes, then the nodes'
  LB_TRY(hipMalloc(&d_visits.p, size_t(n) * 4));
  LB_TRY(hipMemcpy(d_bounds.p, bounds6, size_t(n) * 6 * sizeof(float), hipMemcpyHostToDevice));
  const uint32_t mm_init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  LB_TRY(hipMemcpy(d_mm.p, mm_init, sizeof(mm_init), hipMemcpyHostToDevice));
  LB_TRY(hipMemset(d_visits.p, 0, size_t(n) * 4));

  hipLaunchKernelGGL(lb_centre_bounds, dim3(blocks), dim3(threads), 0, 0, d_bounds.as<float>(), n,
                     d_mm.as<uint32_t>());
  hipLaunchKernelGGL(lb_morton, dim3(blocks), dim3(threads), 0, 0, d_bounds.as<float>(), n,
                     d_mm.as<uint32_t>(), d_keys.as<unsigned long long>());
  size_t tmp_bytes = 0;
  LB_TRY(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_keys.as<unsigned long long>(),
                                  d_keys2.as<unsigned long long>(), n, 0, 62));
  LB_TRY(hipMalloc(&d_tmp.p, std::max<size_t>(tmp_bytes, 16)));
  LB_TRY(rocprim::radix_sort_keys(d_tmp.p, tmp_bytes, d_keys.as<unsigned long long>(),
                                  d_keys2.as<unsigned long long>(), n, 0, 62));
  if (n > 1)
    hipLaunchKernelGGL(lb_radix_tree, dim3(blocks), dim3(threads), 0, 0, d_keys2.as<unsigned long long>(), int(n),
                       d_left.as<uint32_t>(), d_right.as<uint32_t>(), d_pi.as<uint32_t>(), d_pl.as<uint32_t>());
  hipLaunchKernelGGL(lb_boxes, dim3(blocks), dim3(threads), 0, 0, d_keys2.as<unsigned long long>(),
                     d_bounds.as<float>(), int(n), d_left.as<uint32_t>(), d_right.as<uint32_t>(),
                     d_pi.as<uint32_t>(), d_pl.as<uint32_t>(), d_box.as<float>(), d_box.as<float>() + size_t(n) * 6,
                     d_visits.as<uint32_t>());
  LB_TRY(hipGetLastError());
  // (one primitive per leaf: no subtree ends as a leaf; the root is internal node 0, or the only leaf)
  const EmitTree tree{d_left.as<uint32_t>(), d_right.as<uint32_t>(), d_box.as<float>(), nullptr, nullptr, d_keys2.as<unsigned long long>(), n, 1u};
  return emit_reference_layout(tree, n == 1 ? 0u : n, num_nodes, max_depth, nodes, bb, obj_indices);
}


// PLOC builder: same signature and output layout as vimg_hip_build_lbvh; leaves collapsed by the
// SAH as the reference's builders end theirs (up to 8 primitives), the top rebuilt by binned SAH.  The
// tree and its layout are made on the GPU.
extern "C" int vimg_hip_build_ploc(uint32_t n, const float* bounds6, uint32_t* num_nodes,
                                   uint32_t* max_depth, VimgBVHNode* nodes, float* bb,
                                   uint32_t* obj_indices) {
  if (!bounds6 || !num_nodes || !max_depth || !nodes || !bb || !obj_indices || n == 0 || n > (1u << 25))
    return VIMG_E_INVALID;
  if (vimg_hip_device_count() <= 0) return VIMG_E_DEVICE;
  const bool diag = getenv("VIMG_HIP_DIAG") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  const uint32_t threads = 256, blocks = (n + threads - 1) / threads;
  // ids: [0, n) the sorted leaves, [n, 2n - 1) the nodes of the agglomeration, from 2n - 1 the nodes of the new top
  // (at most one per subtree of the cut, and the cut has at most n subtrees)
  const size_t ids = size_t(3) * n;
  Buf d_bounds, d_mm, d_keys, d_keys2, d_tmp, d_box, d_cl[2], d_nn, d_merged, d_keep, d_pos, d_left, d_right, d_counter, d_scan;
  Buf d_nprims, d_cost, d_as_leaf, d_parent, d_firstpos, d_key, d_sorted, d_sort_tmp, d_at, d_items[2], d_jobs[2], d_counts;
  LB_TRY(hipMalloc(&d_bounds.p, size_t(n) * 6 * sizeof(float)));
  LB_TRY(hipMalloc(&d_mm.p, 6 * sizeof(uint32_t)));
  LB_TRY(hipMalloc(&d_keys.p, size_t(n) * 8));
  LB_TRY(hipMalloc(&d_keys2.p, size_t(n) * 8));
  LB_TRY(hipMalloc(&d_box.p, ids * 6 * sizeof(float)));
  for (auto& b : d_cl) LB_TRY(hipMalloc(&b.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_nn.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_merged.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_keep.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_pos.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_left.p, size_t(2) * n * 4));
  LB_TRY(hipMalloc(&d_right.p, size_t(2) * n * 4));
  LB_TRY(hipMalloc(&d_counter.p, 4));
  LB_TRY(hipMalloc(&d_nprims.p, ids * 4));
  LB_TRY(hipMalloc(&d_as_leaf.p, ids * 4));
  LB_TRY(hipMalloc(&d_cost.p, size_t(2) * n * 4));
  LB_TRY(hipMalloc(&d_parent.p, size_t(2) * n * 4));
  LB_TRY(hipMalloc(&d_firstpos.p, size_t(2) * n * 4));
  const PlocRec rec{d_nprims.as<uint32_t>(), d_cost.as<float>(), d_as_leaf.as<uint32_t>(), d_parent.as<uint32_t>(), d_firstpos.as<uint32_t>()};
  struct Ev {   // (destroyed on every way out)
    hipEvent_t e = nullptr;
    ~Ev() { if (e) (void)hipEventDestroy(e); }
  } e0, e1;
  LB_TRY(hipEventCreate(&e0.e));
  LB_TRY(hipEventCreate(&e1.e));
  const hipEvent_t ev0 = e0.e, ev1 = e1.e;
  LB_TRY(hipMemcpy(d_bounds.p, bounds6, size_t(n) * 6 * sizeof(float), hipMemcpyHostToDevice));
  const uint32_t mm_init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  LB_TRY(hipMemcpy(d_mm.p, mm_init, sizeof(mm_init), hipMemcpyHostToDevice));
  LB_TRY(hipMemset(d_counter.p, 0, 4));
  LB_TRY(hipEventRecord(ev0, 0));
  hipLaunchKernelGGL(lb_centre_bounds, dim3(blocks), dim3(threads), 0, 0, d_bounds.as<float>(), n, d_mm.as<uint32_t>());
  hipLaunchKernelGGL(lb_morton, dim3(blocks), dim3(threads), 0, 0, d_bounds.as<float>(), n, d_mm.as<uint32_t>(),
                     d_keys.as<unsigned long long>());
  size_t tmp_bytes = 0, scan_bytes = 0;
  LB_TRY(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(), n, 0, 62));
  LB_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, d_keep.as<uint32_t>(), d_pos.as<uint32_t>(), 0u, n, rocprim::plus<uint32_t>()));
  LB_TRY(hipMalloc(&d_tmp.p, std::max<size_t>(tmp_bytes, 16)));
  LB_TRY(hipMalloc(&d_scan.p, std::max<size_t>(scan_bytes, 16)));
  LB_TRY(rocprim::radix_sort_keys(d_tmp.p, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(), n, 0, 62));
  hipLaunchKernelGGL(ploc_leaf_boxes, dim3(blocks), dim3(threads), 0, 0, d_keys2.as<unsigned long long>(),
                     d_bounds.as<float>(), n, d_box.as<float>(), d_cl[0].as<uint32_t>(), rec);
  uint32_t c = n, radius = PLOC_R;
  if (const char* e = getenv("VIMG_PLOC_R")) radius = uint32_t(std::max(1, atoi(e)));
  int cur = 0, rounds = 0;
  while (c > 1) {
    const uint32_t cb = (c + threads - 1) / threads;
    hipLaunchKernelGGL(ploc_nearest, dim3(cb), dim3(threads), 0, 0, d_cl[cur].as<uint32_t>(), d_box.as<float>(), c,
                       radius, d_nn.as<uint32_t>());
    hipLaunchKernelGGL(ploc_merge, dim3(cb), dim3(threads), 0, 0, d_cl[cur].as<uint32_t>(), d_nn.as<uint32_t>(), c, n,
                       d_box.as<float>(), d_left.as<uint32_t>(), d_right.as<uint32_t>(), d_counter.as<uint32_t>(),
                       d_merged.as<uint32_t>(), d_keep.as<uint32_t>(), rec);
    LB_TRY(rocprim::exclusive_scan(d_scan.p, scan_bytes, d_keep.as<uint32_t>(), d_pos.as<uint32_t>(), 0u, c, rocprim::plus<uint32_t>()));
    hipLaunchKernelGGL(ploc_compact, dim3(cb), dim3(threads), 0, 0, d_merged.as<uint32_t>(), d_keep.as<uint32_t>(),
                       d_pos.as<uint32_t>(), c, d_cl[cur ^ 1].as<uint32_t>());
    uint32_t created = 0;
    LB_TRY(hipMemcpy(&created, d_counter.p, 4, hipMemcpyDeviceToHost));   // (also the round's synchronisation)
    const uint32_t c_new = n - created;
    if (c_new >= c || ++rounds > 4096) return VIMG_E_DEVICE;   // a round always merges the closest pair at least
    c = c_new;
    cur ^= 1;
  }
  LB_TRY(hipGetLastError());
  uint32_t root_id = 0;
  LB_TRY(hipMemcpy(&root_id, d_cl[cur].p, 4, hipMemcpyDeviceToHost));

  // ---- the top, rebuilt over the cut (nothing to do for trees of a few leaves)
  uint32_t top_items_n = 0, top_levels = 0;
  if (n > 16u && !getenv("VIMG_PLOC_NO_TOP")) {
    uint32_t cut = kTopItems;
    if (const char* e = getenv("VIMG_PLOC_TOP")) cut = uint32_t(std::max(4, atoi(e)));
    const uint32_t internal = n - 1u, opened = std::min(cut - 1u, internal);
    const uint32_t nb = (internal + threads - 1) / threads, idb = (2u * n + threads - 1) / threads;
    LB_TRY(hipMalloc(&d_key.p, size_t(n) * 4));
    LB_TRY(hipMalloc(&d_sorted.p, size_t(n) * 4));
    LB_TRY(hipMalloc(&d_at.p, size_t(n) * 4));
    size_t sort_bytes = 0;
    LB_TRY(rocprim::radix_sort_keys_desc(nullptr, sort_bytes, d_key.as<uint32_t>(), d_sorted.as<uint32_t>(), internal));
    LB_TRY(hipMalloc(&d_sort_tmp.p, std::max<size_t>(sort_bytes, 16)));
    hipLaunchKernelGGL(top_keys, dim3(nb), dim3(threads), 0, 0, d_box.as<float>(), d_nprims.as<uint32_t>(), n, d_key.as<uint32_t>());
    LB_TRY(rocprim::radix_sort_keys_desc(d_sort_tmp.p, sort_bytes, d_key.as<uint32_t>(), d_sorted.as<uint32_t>(), internal));
    LB_TRY(hipMemsetAsync(d_keep.p, 0, size_t(n) * 4, 0));   // (the flags of the cut, by first sorted position)
    hipLaunchKernelGGL(top_mark, dim3(idb), dim3(threads), 0, 0, d_key.as<uint32_t>(), d_sorted.as<uint32_t>(), opened, n,
                       d_parent.as<uint32_t>(), d_firstpos.as<uint32_t>(), d_keep.as<uint32_t>(), d_at.as<uint32_t>());
    LB_TRY(rocprim::exclusive_scan(d_scan.p, scan_bytes, d_keep.as<uint32_t>(), d_pos.as<uint32_t>(), 0u, n, rocprim::plus<uint32_t>()));
    uint32_t last_pos = 0, last_flag = 0;
    LB_TRY(hipMemcpy(&last_pos, d_pos.as<uint32_t>() + (n - 1u), 4, hipMemcpyDeviceToHost));
    LB_TRY(hipMemcpy(&last_flag, d_keep.as<uint32_t>() + (n - 1u), 4, hipMemcpyDeviceToHost));
    top_items_n = last_pos + last_flag;
    if (top_items_n >= 3u) {
      for (auto& b : d_items) LB_TRY(hipMalloc(&b.p, size_t(top_items_n) * sizeof(TopItem)));
      for (auto& b : d_jobs) LB_TRY(hipMalloc(&b.p, (size_t(top_items_n) / 2u + 1u) * sizeof(TopJob)));
      uint32_t lg = 0;
      while ((1u << lg) < top_items_n) ++lg;
      top_levels = TOP_SAH_LEVELS + lg + 1u;
      LB_TRY(hipMalloc(&d_counts.p, size_t(top_levels + 1u) * 4));
      LB_TRY(hipMemsetAsync(d_counts.p, 0, size_t(top_levels + 1u) * 4, 0));
      hipLaunchKernelGGL(top_items, dim3(blocks), dim3(threads), 0, 0, d_keep.as<uint32_t>(), d_pos.as<uint32_t>(), d_at.as<uint32_t>(), n,
                         d_box.as<float>(), d_nprims.as<uint32_t>(), d_items[0].as<TopItem>());
      const uint32_t new_root = n + (n - 1u), one = 1u, next = n;   // (the agglomeration made n - 1 nodes)
      const TopJob first{0u, top_items_n, new_root, 0u};
      LB_TRY(hipMemcpy(d_jobs[0].p, &first, sizeof(first), hipMemcpyHostToDevice));
      LB_TRY(hipMemcpy(d_counts.p, &one, 4, hipMemcpyHostToDevice));
      LB_TRY(hipMemcpy(d_counter.p, &next, 4, hipMemcpyHostToDevice));
      for (uint32_t level = 0; level < top_levels; ++level) {
        // (at most 2^level nodes on a level, and every node of a level holds two items at least)
        const uint64_t most = std::min<uint64_t>(level < 31u ? (1ull << level) : ~0ull, top_items_n / 2u);
        hipLaunchKernelGGL(top_level, dim3(uint32_t((most + 3u) / 4u)), dim3(256), 0, 0, d_items[level & 1u].as<TopItem>(),
                           d_items[(level & 1u) ^ 1u].as<TopItem>(), d_jobs[level & 1u].as<TopJob>(), d_jobs[(level & 1u) ^ 1u].as<TopJob>(),
                           d_counts.as<uint32_t>(), level, n, d_counter.as<uint32_t>(), d_box.as<float>(), d_left.as<uint32_t>(),
                           d_right.as<uint32_t>(), d_nprims.as<uint32_t>(), d_as_leaf.as<uint32_t>());
      }
      root_id = new_root;
    }
  }
  LB_TRY(hipEventRecord(ev1, 0));
  LB_TRY(hipGetLastError());
  if (top_levels) {
    uint32_t left_over = 0, made = 0;
    LB_TRY(hipMemcpy(&made, d_counter.p, 4, hipMemcpyDeviceToHost));   // nodes in all (the copy waits for the kernels)
    LB_TRY(hipMemcpy(&left_over, d_counts.as<uint32_t>() + top_levels, 4, hipMemcpyDeviceToHost));
    if (left_over != 0u || made > 2u * n) return VIMG_E_DEVICE;   // (the median levels end every range: not reachable)
  }
  LB_TRY(hipDeviceSynchronize());
  float gpu_ms = 0.f;
  (void)hipEventElapsedTime(&gpu_ms, ev0, ev1);
  const auto t_gpu = std::chrono::steady_clock::now();
  const EmitTree tree{d_left.as<uint32_t>(), d_right.as<uint32_t>(), d_box.as<float>(), d_nprims.as<uint32_t>(), d_as_leaf.as<uint32_t>(),
                      d_keys2.as<unsigned long long>(), n, 0u};
  const int er = emit_reference_layout(tree, root_id, num_nodes, max_depth, nodes, bb, obj_indices);
  if (er != VIMG_OK) return er;
  if (diag) {
    const auto t_end = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "[vimg build] ploc: %u primitives, %d rounds, cut of %u subtrees, %u top levels launched; first kernel to last %.2f ms "
                    "(%.2f ms with the allocations and the upload before), the reference's layout (levels on the GPU) and its download %.2f ms\n",
            n, rounds, top_items_n, top_levels, gpu_ms, ms(t_begin, t_gpu), ms(t_gpu, t_end));
  }
  return VIMG_OK;
}
