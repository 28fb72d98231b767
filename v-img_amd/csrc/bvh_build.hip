// GPU BVH builder: a linear BVH (Morton order + Karras' radix tree +
// bottom-up boxes) emitted in the reference's BVH layout (include/bvh.h:22-57: nodes
// {first_index, obj_count}, BB_mins_maxes with the sibling boxes of a pair side by side,
// obj_indices, max_depth), so that the same upload, the same kernels and the same oracle walk it.
//
// It is NOT the reference's builder: the reference builds a sweep / binned SAH tree on the host
// (src/bvh/sweep_bvh.cpp, bin_bvh.cpp; restated in v-img_amd/host/bvh_build.cpp, which stays the
// default).  An LBVH has one primitive per leaf and no cost model; it is here for scenes whose
// geometry changes between frames, where a 1 M-triangle build in a few milliseconds matters more
// than a 1.5x slower walk.  Parity is unaffected by construction: GPU kernels and oracle read
// whatever tree the scene carries.
//
// Steps (all on the GPU except the last):
//   1. bounds of the primitive centres                       (atomic min / max on ordered ints)
//   2. 30-bit Morton code of each centre, made unique by appending the primitive index
//   3. radix sort of the 64-bit keys                          (rocPRIM)
//   4. Karras 2012: one thread per internal node finds its key range and its split
//   5. boxes bottom-up: the second thread to arrive at a node merges its children
//   6. host: renumber breadth-first so that siblings are adjacent (the layout's rule), fill the
//      sibling-pair box table, count the depth.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
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
  if (i >= n) return;
  const float* b = bounds + size_t(i) * 6;
  for (int a = 0; a < 3; ++a) {
    const uint32_t c = ordered((b[a] + b[3 + a]) * 0.5f);
    atomicMin(&mm[a], c);
    atomicMax(&mm[3 + a], c);
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

// ---- host: the top of a bottom-up tree, rebuilt top-down.  Agglomeration decides well near the
// leaves and poorly near the root (its last merges join whatever is left); a sweep-SAH build decides
// well at the top and is cheap there.  So: cut the tree at the `kTopItems` subtrees of largest
// surface area, and build a surface-area-heuristic tree over those subtrees (cost of a split =
// area x primitives on each side, all three axes swept, as src/bvh/sweep_bvh.cpp:7-49 sweeps).
// New internal nodes are appended to left / right / nodebox; returns the new root.
constexpr size_t kTopItems = 16384;
uint32_t rebuild_top(uint32_t root, std::vector<uint32_t>& left, std::vector<uint32_t>& right,
                     const std::vector<float>& leafbox, std::vector<float>& nodebox, std::vector<uint32_t>& nprims,
                     std::vector<uint8_t>& as_leaf) {
  auto box_of = [&](uint32_t ref) { return (ref & 0x80000000u) ? &leafbox[size_t(ref & 0x7fffffffu) * 6] : &nodebox[size_t(ref) * 6]; };
  auto half_area6 = [](const float* b) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dx * dz + dy * dz;
  };
  auto openable = [&](uint32_t ref) { return !(ref & 0x80000000u) && !as_leaf[ref]; };
  // frontier: always open the subtree of largest area
  std::vector<std::pair<float, uint32_t>> heap;   // (area, ref) of openable subtrees
  std::vector<uint32_t> items;
  auto add = [&](uint32_t ref) {
    if (openable(ref)) {
      heap.push_back({half_area6(box_of(ref)), ref});
      std::push_heap(heap.begin(), heap.end());
    } else {
      items.push_back(ref);
    }
  };
  add(root);
  while (!heap.empty() && heap.size() + items.size() < kTopItems) {
    std::pop_heap(heap.begin(), heap.end());
    const uint32_t r = heap.back().second;
    heap.pop_back();
    add(left[r]);
    add(right[r]);
  }
  for (auto& h : heap) items.push_back(h.second);
  if (items.size() < 3) return root;
  struct It {
    float lo[3], hi[3], c[3];
    uint32_t ref, prims;
  };
  std::vector<It> it(items.size());
  for (size_t i = 0; i < items.size(); ++i) {
    const float* b = box_of(items[i]);
    for (int a = 0; a < 3; ++a) it[i].lo[a] = b[a], it[i].hi[a] = b[3 + a], it[i].c[a] = 0.5f * (b[a] + b[3 + a]);
    it[i].ref = items[i];
    it[i].prims = (items[i] & 0x80000000u) ? 1u : nprims[items[i]];
  }
  std::vector<float> right_cost(items.size());
  // iterative top-down build over ranges of `it`
  struct Job { size_t lo, hi; uint32_t node; };   // node: id of the internal node to fill
  auto new_node = [&]() {
    left.push_back(0), right.push_back(0);
    nodebox.resize(nodebox.size() + 6);
    nprims.push_back(0);
    as_leaf.push_back(0);
    return static_cast<uint32_t>(left.size() - 1);
  };
  const uint32_t new_root = new_node();
  std::vector<Job> jobs{{0, it.size(), new_root}};
  while (!jobs.empty()) {
    const Job j = jobs.back();
    jobs.pop_back();
    const size_t m = j.hi - j.lo;
    // box and primitive count of the range
    float bl[3] = {3.4e38f, 3.4e38f, 3.4e38f}, bh[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    uint32_t np = 0;
    for (size_t i = j.lo; i < j.hi; ++i) {
      for (int a = 0; a < 3; ++a) bl[a] = std::min(bl[a], it[i].lo[a]), bh[a] = std::max(bh[a], it[i].hi[a]);
      np += it[i].prims;
    }
    for (int a = 0; a < 3; ++a) nodebox[size_t(j.node) * 6 + a] = bl[a], nodebox[size_t(j.node) * 6 + 3 + a] = bh[a];
    nprims[j.node] = np;
    // best split over the three axes
    float best = 3.4e38f;
    int best_axis = 0;
    size_t best_k = j.lo + m / 2;
    for (int a = 0; a < 3; ++a) {
      std::sort(it.begin() + j.lo, it.begin() + j.hi, [a](const It& x, const It& y) { return x.c[a] < y.c[a]; });
      float l[3] = {3.4e38f, 3.4e38f, 3.4e38f}, h[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
      uint32_t cnt = 0;
      for (size_t i = j.hi; i-- > j.lo + 1;) {   // suffix costs
        for (int x = 0; x < 3; ++x) l[x] = std::min(l[x], it[i].lo[x]), h[x] = std::max(h[x], it[i].hi[x]);
        cnt += it[i].prims;
        const float b6[6] = {l[0], l[1], l[2], h[0], h[1], h[2]};
        right_cost[i] = half_area6(b6) * float(cnt);
      }
      float pl[3] = {3.4e38f, 3.4e38f, 3.4e38f}, ph[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
      uint32_t lc = 0;
      for (size_t i = j.lo; i + 1 < j.hi; ++i) {
        for (int x = 0; x < 3; ++x) pl[x] = std::min(pl[x], it[i].lo[x]), ph[x] = std::max(ph[x], it[i].hi[x]);
        lc += it[i].prims;
        const float b6[6] = {pl[0], pl[1], pl[2], ph[0], ph[1], ph[2]};
        const float cost = half_area6(b6) * float(lc) + right_cost[i + 1];
        if (cost < best) best = cost, best_axis = a, best_k = i + 1;
      }
    }
    if (best_axis != 2)
      std::sort(it.begin() + j.lo, it.begin() + j.hi, [best_axis](const It& x, const It& y) { return x.c[best_axis] < y.c[best_axis]; });
    const size_t k = best_k;
    auto child = [&](size_t lo, size_t hi) -> uint32_t {
      if (hi - lo == 1) return it[lo].ref;
      const uint32_t id = new_node();
      jobs.push_back({lo, hi, id});
      return id;
    };
    const uint32_t lref = child(j.lo, k), rref = child(k, j.hi);
    left[j.node] = lref;
    right[j.node] = rref;
  }
  return new_root;
}

// ---- host: a binary tree over the sorted leaves (child reference with bit 31 set = leaf at that
// sorted position, else an internal node id) -> the reference layout (include/bvh.h:22-57):
// breadth-first numbering with the two children of a node adjacent, the sibling-pair box table,
// obj_indices with every leaf's primitives contiguous, max_depth.  `collapse`: subtrees of at most
// 8 primitives become ONE leaf where the surface-area heuristic of the reference's builders
// (traversal 0.5, intersection 1: include/bvh.h:17-20, src/bvh/sweep_bvh.cpp:140-147) says a leaf
// is no dearer than the split - which is how the reference's own trees end (leaves of up to 8).
void emit_reference_layout(uint32_t n, uint32_t root_ref, std::vector<uint32_t> left,
                           std::vector<uint32_t> right, const std::vector<float>& leafbox,
                           std::vector<float> nodebox, const std::vector<uint32_t>& sorted_prim, bool collapse,
                           uint32_t* num_nodes, uint32_t* max_depth, VimgBVHNode* nodes, float* bb,
                           uint32_t* obj_indices) {
  auto box_of = [&](uint32_t ref) { return (ref & 0x80000000u) ? &leafbox[size_t(ref & 0x7fffffffu) * 6] : &nodebox[size_t(ref) * 6]; };
  auto half_area = [&](const float* b) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dx * dz + dy * dz;
  };
  auto put = [&](size_t bb_index, const float* v) { std::memcpy(bb + bb_index * 3, v, 12); };
  // bottom-up: primitives and SAH cost of every internal node (explicit stack: LBVH chains can be long)
  std::vector<uint32_t> nprims(n, 1u);
  std::vector<float> cost(n, 1.f);
  std::vector<uint8_t> as_leaf(n, 0);
  if (!(root_ref & 0x80000000u)) {
    std::vector<std::pair<uint32_t, uint8_t>> st;
    st.push_back({root_ref, 0});
    while (!st.empty()) {
      auto [node, state] = st.back();
      if (state == 0) {
        st.back().second = 1;
        if (!(left[node] & 0x80000000u)) st.push_back({left[node], 0});
        if (!(right[node] & 0x80000000u)) st.push_back({right[node], 0});
      } else {
        st.pop_back();
        const uint32_t c[2] = {left[node], right[node]};
        uint32_t np = 0;
        float split = 0.5f;
        const float area = half_area(box_of(node));
        for (int k = 0; k < 2; ++k) {
          const bool lf = (c[k] & 0x80000000u) != 0;
          np += lf ? 1u : nprims[c[k]];
          const float cc = lf ? 1.f : cost[c[k]];
          split += (area > 0.f ? half_area(box_of(c[k])) / area : 1.f) * cc;
        }
        nprims[node] = np;
        const float leaf = 1.f * float(np);
        as_leaf[node] = collapse && np <= 8u && leaf <= split;
        cost[node] = as_leaf[node] ? leaf : split;
      }
    }
  }
  if (collapse && !(root_ref & 0x80000000u) && !as_leaf[root_ref])
    root_ref = rebuild_top(root_ref, left, right, leafbox, nodebox, nprims, as_leaf);
  struct Item { uint32_t ref, out, depth; };
  std::vector<Item> queue;
  queue.reserve(size_t(n) * 2);
  queue.push_back({root_ref, 0u, 1u});
  put(0, box_of(root_ref));
  put(2, box_of(root_ref) + 3);
  uint32_t next = 1, deepest = 1, cursor = 0;
  std::vector<uint32_t> dfs;
  for (size_t head = 0; head < queue.size(); ++head) {
    const Item it = queue[head];
    deepest = std::max(deepest, it.depth);
    if (it.ref & 0x80000000u) {
      nodes[it.out] = VimgBVHNode{cursor, 1u};
      obj_indices[cursor++] = sorted_prim[it.ref & 0x7fffffffu];
      continue;
    }
    if (as_leaf[it.ref]) {
      nodes[it.out] = VimgBVHNode{cursor, nprims[it.ref]};
      dfs.assign(1, it.ref);
      while (!dfs.empty()) {
        const uint32_t r = dfs.back();
        dfs.pop_back();
        if (r & 0x80000000u) {
          obj_indices[cursor++] = sorted_prim[r & 0x7fffffffu];
        } else {
          dfs.push_back(right[r]);
          dfs.push_back(left[r]);
        }
      }
      continue;
    }
    const uint32_t first_child = next;
    next += 2;
    nodes[it.out] = VimgBVHNode{first_child, 0u};
    const uint32_t c[2] = {left[it.ref], right[it.ref]};
    const size_t base = size_t(first_child) * 2 + 2;   // {Lmin, Rmin, Lmax, Rmax}
    put(base + 0, box_of(c[0]));
    put(base + 1, box_of(c[1]));
    put(base + 2, box_of(c[0]) + 3);
    put(base + 3, box_of(c[1]) + 3);
    queue.push_back({c[0], first_child, it.depth + 1});
    queue.push_back({c[1], first_child + 1, it.depth + 1});
  }
  *num_nodes = next;
  *max_depth = deepest;
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
__global__ void ploc_leaf_boxes(const unsigned long long* __restrict__ keys, const float* __restrict__ bounds, uint32_t n,
                                float* __restrict__ box, uint32_t* __restrict__ cluster) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t prim = static_cast<uint32_t>(keys[k] & 0xffffffffull);
  for (int a = 0; a < 6; ++a) box[size_t(k) * 6 + a] = bounds[size_t(prim) * 6 + a];
  cluster[k] = k;
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
                           uint32_t* __restrict__ next_node, uint32_t* __restrict__ merged, uint32_t* __restrict__ keep) {
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
      for (int x = 0; x < 3; ++x) {
        box[size_t(id) * 6 + x] = fminf(box[size_t(a) * 6 + x], box[size_t(b) * 6 + x]);
        box[size_t(id) * 6 + 3 + x] = fmaxf(box[size_t(a) * 6 + 3 + x], box[size_t(b) * 6 + 3 + x]);
      }
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

struct Buf {
  void* p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
  template <typename T> T* as() { return static_cast<T*>(p); }
};

}  // namespace

extern "C" int vimg_hip_build_lbvh(uint32_t n, const float* bounds6, uint32_t* num_nodes,
                                   uint32_t* max_depth, VimgBVHNode* nodes, float* bb,
                                   uint32_t* obj_indices) {
  if (!bounds6 || !num_nodes || !max_depth || !nodes || !bb || !obj_indices || n == 0 || n > (1u << 25))
    return VIMG_E_INVALID;
  if (vimg_hip_device_count() <= 0) return VIMG_E_DEVICE;
  const uint32_t threads = 256, blocks = (n + threads - 1) / threads;
  Buf d_bounds, d_mm, d_keys, d_keys2, d_left, d_right, d_pi, d_pl, d_leafbox, d_nodebox, d_visits, d_tmp;
  LB_TRY(hipMalloc(&d_bounds.p, size_t(n) * 6 * sizeof(float)));
  LB_TRY(hipMalloc(&d_mm.p, 6 * sizeof(uint32_t)));
  LB_TRY(hipMalloc(&d_keys.p, size_t(n) * 8));
  LB_TRY(hipMalloc(&d_keys2.p, size_t(n) * 8));
  LB_TRY(hipMalloc(&d_left.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_right.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_pi.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_pl.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_leafbox.p, size_t(n) * 6 * sizeof(float)));
  LB_TRY(hipMalloc(&d_nodebox.p, size_t(n) * 6 * sizeof(float)));
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
                     d_pi.as<uint32_t>(), d_pl.as<uint32_t>(), d_leafbox.as<float>(), d_nodebox.as<float>(),
                     d_visits.as<uint32_t>());
  LB_TRY(hipGetLastError());
  LB_TRY(hipDeviceSynchronize());

  std::vector<unsigned long long> keys(n);
  std::vector<uint32_t> left(n), right(n);
  std::vector<float> leafbox(size_t(n) * 6), nodebox(size_t(n) * 6);
  LB_TRY(hipMemcpy(keys.data(), d_keys2.p, size_t(n) * 8, hipMemcpyDeviceToHost));
  LB_TRY(hipMemcpy(leafbox.data(), d_leafbox.p, size_t(n) * 6 * sizeof(float), hipMemcpyDeviceToHost));
  if (n > 1) {
    LB_TRY(hipMemcpy(left.data(), d_left.p, size_t(n) * 4, hipMemcpyDeviceToHost));
    LB_TRY(hipMemcpy(right.data(), d_right.p, size_t(n) * 4, hipMemcpyDeviceToHost));
    LB_TRY(hipMemcpy(nodebox.data(), d_nodebox.p, size_t(n) * 6 * sizeof(float), hipMemcpyDeviceToHost));
  }
  std::vector<uint32_t> sorted_prim(n);
  for (uint32_t k = 0; k < n; ++k) sorted_prim[k] = static_cast<uint32_t>(keys[k] & 0xffffffffull);
  emit_reference_layout(n, (n == 1) ? 0x80000000u : 0u, left, right, leafbox, nodebox, sorted_prim, false,
                        num_nodes, max_depth, nodes, bb, obj_indices);
  return VIMG_OK;
}


// PLOC builder: same signature and output layout as vimg_hip_build_lbvh; leaves collapsed by the
// SAH as the reference's builders end theirs (up to 8 primitives).
extern "C" int vimg_hip_build_ploc(uint32_t n, const float* bounds6, uint32_t* num_nodes,
                                   uint32_t* max_depth, VimgBVHNode* nodes, float* bb,
                                   uint32_t* obj_indices) {
  if (!bounds6 || !num_nodes || !max_depth || !nodes || !bb || !obj_indices || n == 0 || n > (1u << 25))
    return VIMG_E_INVALID;
  if (vimg_hip_device_count() <= 0) return VIMG_E_DEVICE;
  const uint32_t threads = 256, blocks = (n + threads - 1) / threads;
  Buf d_bounds, d_mm, d_keys, d_keys2, d_tmp, d_box, d_cl[2], d_nn, d_merged, d_keep, d_pos, d_left, d_right, d_counter, d_scan;
  LB_TRY(hipMalloc(&d_bounds.p, size_t(n) * 6 * sizeof(float)));
  LB_TRY(hipMalloc(&d_mm.p, 6 * sizeof(uint32_t)));
  LB_TRY(hipMalloc(&d_keys.p, size_t(n) * 8));
  LB_TRY(hipMalloc(&d_keys2.p, size_t(n) * 8));
  LB_TRY(hipMalloc(&d_box.p, size_t(2) * n * 6 * sizeof(float)));
  for (auto& b : d_cl) LB_TRY(hipMalloc(&b.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_nn.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_merged.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_keep.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_pos.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_left.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_right.p, size_t(n) * 4));
  LB_TRY(hipMalloc(&d_counter.p, 4));
  LB_TRY(hipMemcpy(d_bounds.p, bounds6, size_t(n) * 6 * sizeof(float), hipMemcpyHostToDevice));
  const uint32_t mm_init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  LB_TRY(hipMemcpy(d_mm.p, mm_init, sizeof(mm_init), hipMemcpyHostToDevice));
  LB_TRY(hipMemset(d_counter.p, 0, 4));
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
                     d_bounds.as<float>(), n, d_box.as<float>(), d_cl[0].as<uint32_t>());
  uint32_t c = n, radius = PLOC_R;
  if (const char* e = getenv("VIMG_PLOC_R")) radius = uint32_t(std::max(1, atoi(e)));
  int cur = 0, rounds = 0;
  while (c > 1) {
    const uint32_t cb = (c + threads - 1) / threads;
    hipLaunchKernelGGL(ploc_nearest, dim3(cb), dim3(threads), 0, 0, d_cl[cur].as<uint32_t>(), d_box.as<float>(), c,
                       radius, d_nn.as<uint32_t>());
    hipLaunchKernelGGL(ploc_merge, dim3(cb), dim3(threads), 0, 0, d_cl[cur].as<uint32_t>(), d_nn.as<uint32_t>(), c, n,
                       d_box.as<float>(), d_left.as<uint32_t>(), d_right.as<uint32_t>(), d_counter.as<uint32_t>(),
                       d_merged.as<uint32_t>(), d_keep.as<uint32_t>());
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
  std::vector<unsigned long long> keys(n);
  std::vector<uint32_t> left(n), right(n), root(1, 0);
  std::vector<float> box(size_t(2) * n * 6);
  LB_TRY(hipMemcpy(keys.data(), d_keys2.p, size_t(n) * 8, hipMemcpyDeviceToHost));
  LB_TRY(hipMemcpy(box.data(), d_box.p, box.size() * sizeof(float), hipMemcpyDeviceToHost));
  LB_TRY(hipMemcpy(left.data(), d_left.p, size_t(n) * 4, hipMemcpyDeviceToHost));
  LB_TRY(hipMemcpy(right.data(), d_right.p, size_t(n) * 4, hipMemcpyDeviceToHost));
  LB_TRY(hipMemcpy(root.data(), d_cl[cur].p, 4, hipMemcpyDeviceToHost));
  // to the emitter's form: internal ids from 0, leaf references with bit 31
  auto ref = [&](uint32_t id) { return id < n ? (0x80000000u | id) : id - n; };
  for (uint32_t k = 0; k + 1 < n; ++k) left[k] = ref(left[k]), right[k] = ref(right[k]);
  std::vector<float> leafbox(box.begin(), box.begin() + size_t(n) * 6), nodebox(box.begin() + size_t(n) * 6, box.end());
  std::vector<uint32_t> sorted_prim(n);
  for (uint32_t k = 0; k < n; ++k) sorted_prim[k] = static_cast<uint32_t>(keys[k] & 0xffffffffull);
  emit_reference_layout(n, ref(root[0]), left, right, leafbox, nodebox, sorted_prim, true, num_nodes, max_depth, nodes, bb,
                        obj_indices);
  return VIMG_OK;
}
