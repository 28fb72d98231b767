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

  // ---- the reference layout: breadth-first numbering, the two children of a node adjacent;
  // leaf k of the sorted order owns obj_indices[k]
  for (uint32_t k = 0; k < n; ++k) obj_indices[k] = static_cast<uint32_t>(keys[k] & 0xffffffffull);
  auto box_of = [&](uint32_t ref) { return (ref & 0x80000000u) ? &leafbox[size_t(ref & 0x7fffffffu) * 6] : &nodebox[size_t(ref) * 6]; };
  auto put = [&](size_t bb_index, const float* v) { std::memcpy(bb + bb_index * 3, v, 12); };
  const uint32_t root_ref = (n == 1) ? 0x80000000u : 0u;
  struct Item { uint32_t ref, out, depth; };
  std::vector<Item> queue;
  queue.reserve(size_t(n) * 2);
  queue.push_back({root_ref, 0u, 1u});
  put(0, box_of(root_ref));
  put(2, box_of(root_ref) + 3);
  uint32_t next = 1, deepest = 1;
  for (size_t head = 0; head < queue.size(); ++head) {
    const Item it = queue[head];
    deepest = std::max(deepest, it.depth);
    if (it.ref & 0x80000000u) {
      nodes[it.out] = VimgBVHNode{it.ref & 0x7fffffffu, 1u};
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
  return VIMG_OK;
}
