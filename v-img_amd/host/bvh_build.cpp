// Host SAH BVH builders emitting the reference's flat layout (nodes / sibling-pair AABBs /
// obj_indices).  The north star keeps the build on the CPU; what matters for the hot path is that
// the OUTPUT matches what the reference's builders hand to BVH::hit:
//   sweep : BVH::build_sweep_bvh   reference src/bvh/sweep_bvh.cpp:7-292
//   binned: BVH::build_bin_bvh     reference src/bvh/bin_bvh.cpp:8-235
// Both are written single-threaded and depth-first (left subtree numbered before right), which is
// the node numbering the reference produces whenever it does not spawn build threads (< 1024
// primitives on the left side); above that the reference's numbering is itself non-deterministic
// (atomic node counter), the topology is not.
#include <algorithm>
#include <limits>
#include <numeric>

#include "host_scene.hpp"

using hm::V3;

namespace {

constexpr float kIntersectionCost = 1.f;  // BVHConst, reference include/bvh.h:17-20
constexpr float kTraversalCost = 0.5f;

struct Box {
  V3 lo{+std::numeric_limits<float>::max(), +std::numeric_limits<float>::max(),
        +std::numeric_limits<float>::max()};
  V3 hi{-std::numeric_limits<float>::max(), -std::numeric_limits<float>::max(),
        -std::numeric_limits<float>::max()};
  void grow(const PrimBounds& b) {
    lo = hm::vmin(lo, b.bmin);
    hi = hm::vmax(hi, b.bmax);
  }
  void grow(const Box& b) {
    lo = hm::vmin(lo, b.lo);
    hi = hm::vmax(hi, b.hi);
  }
  float half_area() const {
    V3 d = hi - lo;
    return d.x * d.y + d.x * d.z + d.y * d.z;
  }
  float area() const {
    V3 d = hi - lo;
    return 2.f * (d.x * d.y + d.x * d.z + d.y * d.z);
  }
  uint32_t widest_axis() const {
    V3 d = hi - lo;
    uint32_t a = 0;
    if (d[a] < d[1]) a = 1;
    if (d[a] < d[2]) a = 2;
    return a;
  }
};

Box box_of(const PrimBounds& b) {
  Box r;
  r.lo = b.bmin;
  r.hi = b.bmax;
  return r;
}

struct Candidate {
  size_t first_right;  // number of primitives going left
  float cost;
  uint32_t axis;
  Box left_box;
};

void store_child_boxes(HostBVH& bvh, size_t first_child, const Box& l, const Box& r) {
  size_t base = (first_child * 2 + 2) * 3;
  for (int a = 0; a < 3; ++a) {
    bvh.bb[base + 0 + a] = l.lo[a];
    bvh.bb[base + 3 + a] = r.lo[a];
    bvh.bb[base + 6 + a] = l.hi[a];
    bvh.bb[base + 9 + a] = r.hi[a];
  }
}

Box load_box(const HostBVH& bvh, size_t bb_index) {
  Box b;
  for (int a = 0; a < 3; ++a) {
    b.lo[a] = bvh.bb[bb_index * 3 + a];
    b.hi[a] = bvh.bb[(bb_index + 2) * 3 + a];
  }
  return b;
}

// ---------------------------------------------------------------- sweep SAH

struct SweepCtx {
  HostBVH& bvh;
  const std::vector<PrimBounds>& bboxes;
  size_t next_node;
  uint32_t max_node_prims;
  std::vector<uint8_t> goes_left;
  std::vector<float> right_cost;
};

// Best split of one axis order: candidate i+1 puts sorted[0..i] left.  Strictly-better wins and
// the running best is carried across the three axes, as the reference does; its early-outs only
// skip candidates that cannot be strictly better, so a plain sweep with those bounds is equal.
Candidate sweep_axis(SweepCtx& c, uint32_t axis, const size_t* sorted, size_t n, Candidate best) {
  size_t start = 0;
  Box acc;
  size_t count = 0;
  for (size_t i = n - 1; i > 0; --i) {
    acc.grow(c.bboxes[sorted[i]]);
    ++count;
    c.right_cost[i] = acc.area() * count;
    if (c.right_cost[i] > best.cost) {
      start = i - 1;
      break;
    }
  }
  Box left;
  size_t left_n = 0;
  for (size_t i = 0; i < start; ++i) {
    left.grow(c.bboxes[sorted[i]]);
    ++left_n;
  }
  for (size_t i = start; i + 1 < n; ++i) {
    left.grow(c.bboxes[sorted[i]]);
    ++left_n;
    float lc = left.area() * left_n;
    float total = lc + c.right_cost[i + 1];
    if (total < best.cost) {
      best = Candidate{i + 1, total, axis, left};
    } else if (lc > best.cost) {
      break;
    }
  }
  return best;
}

// order[k] points at n primitive ids sorted along axis k (k = 3 is scratch of the same length).
uint32_t sweep_node(SweepCtx& c, size_t node, size_t bb_index, size_t* order[4], size_t n) {
  VimgBVHNode& cur = c.bvh.nodes[node];
  if (n == 1) {
    c.bvh.obj_indices[cur.first_index] = static_cast<uint32_t>(order[0][0]);
    return 1;
  }
  Box node_box = load_box(c.bvh, bb_index);
  const float leaf_cost = ((kIntersectionCost * n) - kTraversalCost) * node_box.area();
  Candidate best{n / 2, leaf_cost, 0, Box{}};
  for (uint32_t a = 0; a < 3; ++a) best = sweep_axis(c, a, order[a], n, best);

  bool reuse_left_box = true;
  if (best.cost >= leaf_cost) {
    reuse_left_box = false;
    if (n > c.max_node_prims) {
      best.axis = node_box.widest_axis();
      best.first_right = n / 2;
    } else {
      for (size_t i = 0; i < n; ++i)
        c.bvh.obj_indices[cur.first_index + i] = static_cast<uint32_t>(order[0][i]);
      return 1;
    }
  }

  const size_t nl = best.first_right, nr = n - nl;
  for (size_t i = nl; i < n; ++i) c.goes_left[order[best.axis][i]] = 0;
  for (size_t i = 0; i < nl; ++i) c.goes_left[order[best.axis][i]] = 1;
  for (uint32_t a = 0; a < 3; ++a) {
    if (a == best.axis) continue;
    size_t l = 0, r = 0;
    for (size_t i = 0; i < n; ++i) {
      size_t p = order[a][i];
      if (c.goes_left[p])
        order[3][l++] = p;
      else
        order[3][nl + r++] = p;
    }
    std::swap(order[3], order[a]);
  }

  size_t* lo_order[4] = {order[0], order[1], order[2], order[3]};
  size_t* hi_order[4] = {order[0] + nl, order[1] + nl, order[2] + nl, order[3] + nl};
  size_t lo_n = nl, hi_n = nr;

  const size_t first_child = c.next_node;
  c.next_node += 2;
  VimgBVHNode a_node{cur.first_index, static_cast<uint32_t>(nl)};
  VimgBVHNode b_node{cur.first_index + static_cast<uint32_t>(nl), static_cast<uint32_t>(nr)};

  Box a_box;
  if (reuse_left_box) {
    a_box = best.left_box;
  } else {
    a_box = box_of(c.bboxes[lo_order[0][0]]);
    for (size_t i = 1; i < nl; ++i) a_box.grow(c.bboxes[lo_order[0][i]]);
  }
  Box b_box = box_of(c.bboxes[hi_order[0][0]]);
  for (size_t i = 1; i < nr; ++i) b_box.grow(c.bboxes[hi_order[0][i]]);

  // the larger box becomes the second sibling
  if (a_box.half_area() > b_box.half_area()) {
    std::swap(a_box, b_box);
    std::swap(a_node, b_node);
    for (int k = 0; k < 4; ++k) std::swap(lo_order[k], hi_order[k]);
    std::swap(lo_n, hi_n);
  }
  c.bvh.nodes[first_child] = a_node;
  c.bvh.nodes[first_child + 1] = b_node;
  store_child_boxes(c.bvh, first_child, a_box, b_box);
  c.bvh.nodes[node].first_index = static_cast<uint32_t>(first_child);
  c.bvh.nodes[node].obj_count = 0;

  uint32_t d0 = sweep_node(c, first_child, first_child * 2 + 2, lo_order, lo_n);
  uint32_t d1 = sweep_node(c, first_child + 1, first_child * 2 + 3, hi_order, hi_n);
  return 1 + (d0 > d1 ? d0 : d1);
}

// ---------------------------------------------------------------- binned SAH

size_t bin_of(uint32_t axis, const Box& box, const V3& center, size_t num_bins) {
  int index = (center[axis] - box.lo[axis]) * (num_bins / (box.hi[axis] - box.lo[axis]));
  return std::min(num_bins - 1, static_cast<size_t>(std::max(0, index)));
}

struct BinCtx {
  HostBVH& bvh;
  const std::vector<PrimBounds>& bboxes;
  const std::vector<V3>& centers;
  size_t num_bins;
  size_t next_node;
};

uint32_t bin_node(BinCtx& c, size_t node, size_t bb_index) {
  VimgBVHNode cur = c.bvh.nodes[node];
  Box node_box = load_box(c.bvh, bb_index);
  const size_t nb = c.num_bins;

  size_t best_split = std::numeric_limits<size_t>::max();
  float best_cost = std::numeric_limits<float>::max();
  uint32_t best_axis = 0;
  for (uint32_t axis = 0; axis < 3; ++axis) {
    std::vector<Box> bin_box(nb);
    std::vector<size_t> bin_n(nb, 0);
    for (uint32_t i = 0; i < cur.obj_count; ++i) {
      uint32_t p = c.bvh.obj_indices[cur.first_index + i];
      size_t b = bin_of(axis, node_box, c.centers[p], nb);
      bin_box[b].grow(c.bboxes[p]);
      bin_n[b]++;
    }
    std::vector<float> right_cost(nb, 0.f);
    Box acc;
    size_t cnt = 0;
    for (size_t i = nb - 1; i > 0; --i) {
      acc.grow(bin_box[i]);
      cnt += bin_n[i];
      right_cost[i] = acc.area() * cnt;  // empty side: inf * 0 = NaN, ignored by the < below
    }
    Box left;
    size_t ln = 0;
    for (size_t i = 0; i + 1 < nb; ++i) {
      left.grow(bin_box[i]);
      ln += bin_n[i];
      float cost = left.area() * ln + right_cost[i + 1];
      if (cost < best_cost) {
        best_cost = cost;
        best_split = i + 1;
        best_axis = axis;
      }
    }
  }

  float leaf_cost = kIntersectionCost * cur.obj_count;
  float split_cost = kTraversalCost + best_cost / node_box.area();
  size_t first_right;
  auto begin = c.bvh.obj_indices.begin() + cur.first_index;
  auto end = begin + cur.obj_count;
  if (static_cast<float>(best_split) == std::numeric_limits<float>::max() ||
      split_cost >= leaf_cost) {
    if (cur.obj_count > 8) {
      uint32_t axis = node_box.widest_axis();
      std::sort(begin, end,
                [&](uint32_t i, uint32_t j) { return c.centers[i][axis] < c.centers[j][axis]; });
      first_right = cur.first_index + cur.obj_count / 2;
    } else {
      return 1;
    }
  } else {
    first_right = std::partition(begin, end,
                                 [&](uint32_t i) {
                                   return bin_of(best_axis, node_box, c.centers[i], nb) <
                                          best_split;
                                 }) -
                  c.bvh.obj_indices.begin();
  }

  const size_t first_child = c.next_node;
  c.next_node += 2;
  VimgBVHNode a_node{cur.first_index, static_cast<uint32_t>(first_right - cur.first_index)};
  VimgBVHNode b_node{static_cast<uint32_t>(first_right), cur.obj_count - a_node.obj_count};
  Box a_box = box_of(c.bboxes[c.bvh.obj_indices[a_node.first_index]]);
  for (uint32_t i = 1; i < a_node.obj_count; ++i)
    a_box.grow(c.bboxes[c.bvh.obj_indices[a_node.first_index + i]]);
  Box b_box = box_of(c.bboxes[c.bvh.obj_indices[b_node.first_index]]);
  for (uint32_t i = 1; i < b_node.obj_count; ++i)
    b_box.grow(c.bboxes[c.bvh.obj_indices[b_node.first_index + i]]);
  if (a_box.half_area() > b_box.half_area()) {
    std::swap(a_box, b_box);
    std::swap(a_node, b_node);
  }
  c.bvh.nodes[first_child] = a_node;
  c.bvh.nodes[first_child + 1] = b_node;
  store_child_boxes(c.bvh, first_child, a_box, b_box);
  c.bvh.nodes[node].first_index = static_cast<uint32_t>(first_child);
  c.bvh.nodes[node].obj_count = 0;

  uint32_t d0 = bin_node(c, first_child, first_child * 2 + 2);
  uint32_t d1 = bin_node(c, first_child + 1, first_child * 2 + 3);
  return 1 + (d0 > d1 ? d0 : d1);
}

void init_root(HostBVH& bvh, const std::vector<PrimBounds>& bboxes, size_t n) {
  bvh.nodes.assign(2 * n - 1, VimgBVHNode{0, 0});
  bvh.bb.assign(((2 * n - 1) * 2 + 3) * 3, 0.f);
  bvh.nodes[0] = VimgBVHNode{0, static_cast<uint32_t>(n)};
  Box root = box_of(bboxes[bvh.obj_indices[0]]);
  for (size_t i = 1; i < n; ++i) root.grow(bboxes[bvh.obj_indices[i]]);
  for (int a = 0; a < 3; ++a) {
    bvh.bb[0 * 3 + a] = root.lo[a];
    bvh.bb[2 * 3 + a] = root.hi[a];
  }
}

}  // namespace

// setup_for_bvh, reference src/main.cpp:26-36 with Triangle::bounds (src/geometry/triangle.cpp:155)
// and Sphere::bounds (src/geometry/sphere.cpp:47-54)
void prim_bounds(const VimgHostScene& s, std::vector<PrimBounds>& bounds,
                 std::vector<V3>& centers) {
  bounds.clear();
  centers.clear();
  for (const VimgPrim& p : s.prims) {
    PrimBounds b;
    if (p.type == VIMG_PRIM_TRIANGLE) {
      const VimgMesh& m = s.meshes[s.tri_mesh[p.index]];
      V3 v[3];
      for (int k = 0; k < 3; ++k) {
        size_t vi = m.first_vertex + s.tri_indices[p.index * 3 + k];
        v[k] = V3{s.vertices[vi * 3], s.vertices[vi * 3 + 1], s.vertices[vi * 3 + 2]};
      }
      b.bmin = hm::vmin(v[0], hm::vmin(v[1], v[2]));
      b.bmax = hm::vmax(v[0], hm::vmax(v[1], v[2]));
    } else {
      const VimgSphere& sp = s.spheres[p.index];
      V3 c{sp.center[0], sp.center[1], sp.center[2]};
      V3 d{sp.radius, sp.radius, sp.radius};
      b.bmin = c - d;
      b.bmax = c + d;
    }
    bounds.push_back(b);
    centers.push_back((b.bmax + b.bmin) * 0.5f);
  }
}

HostBVH build_sweep_bvh(const std::vector<PrimBounds>& bboxes, const std::vector<V3>& centers,
                        uint32_t max_node_prims) {
  HostBVH bvh;
  const size_t n = bboxes.size();
  if (n == 0) return bvh;
  bvh.obj_indices.resize(n);
  std::iota(bvh.obj_indices.begin(), bvh.obj_indices.end(), 0u);

  std::vector<size_t> sorted[4];
  for (int a = 0; a < 4; ++a) {
    sorted[a].resize(n);
    std::iota(sorted[a].begin(), sorted[a].end(), size_t{0});
  }
  for (int a = 0; a < 3; ++a)
    std::sort(sorted[a].begin(), sorted[a].end(),
              [&](size_t i, size_t j) { return centers[i][a] < centers[j][a]; });

  init_root(bvh, bboxes, n);
  SweepCtx ctx{bvh, bboxes, 1, max_node_prims, std::vector<uint8_t>(n, 0),
               std::vector<float>(n + 1, 0.f)};
  size_t* order[4] = {sorted[0].data(), sorted[1].data(), sorted[2].data(), sorted[3].data()};
  bvh.max_depth = sweep_node(ctx, 0, 0, order, n);
  bvh.nodes.resize(ctx.next_node);
  bvh.bb.resize((ctx.next_node * 2 + 3) * 3);
  return bvh;
}

HostBVH build_bin_bvh(const std::vector<PrimBounds>& bboxes, const std::vector<V3>& centers,
                      size_t num_bins) {
  HostBVH bvh;
  const size_t n = bboxes.size();
  if (n == 0) return bvh;
  bvh.obj_indices.resize(n);
  std::iota(bvh.obj_indices.begin(), bvh.obj_indices.end(), 0u);
  init_root(bvh, bboxes, n);
  BinCtx ctx{bvh, bboxes, centers, num_bins, 1};
  bvh.max_depth = bin_node(ctx, 0, 0);
  bvh.nodes.resize(ctx.next_node);
  bvh.bb.resize((ctx.next_node * 2 + 3) * 3);
  return bvh;
}
