// heatmap_img on the GPU: the traversal-cost picture of the BVH.
//   heatmap_img        reference src/integrators/heatmap.cpp:38-147
//   BVH::hit<float>    reference include/bvh.h:83-225 (scalar path: root test 0.5, each pair of
//                      sibling boxes 2 x 0.5, each primitive test 1)
//   turbo_colormap     reference src/integrators/heatmap.cpp:21-36
// One lane per pixel, all samples of the pixel in sequence (one PCG stream per pixel, as the path
// tracer); the walk is the closest-hit traverse<false> of render_kernels.h with its event
// counters on, so the cost is counted on exactly the node visits and primitive tests the path
// tracer makes.  Costs are multiples of 0.5 far below 2^23: the float sums are exact, so
// 0.5 + visits + tests equals the reference's running sum whatever the order.
#pragma once
#include "render_kernels.h"

namespace vimg {

VD f3 turbo_colormap(float x) {
  // glm::vec4 / vec2 constants are written as double literals in the reference
  const float kR4[4] = {static_cast<float>(0.13572138), static_cast<float>(4.61539260),
                        static_cast<float>(-42.66032258), static_cast<float>(132.13108234)};
  const float kG4[4] = {static_cast<float>(0.09140261), static_cast<float>(2.19418839),
                        static_cast<float>(4.84296658), static_cast<float>(-14.18503333)};
  const float kB4[4] = {static_cast<float>(0.10667330), static_cast<float>(12.64194608),
                        static_cast<float>(-60.58204836), static_cast<float>(110.36276771)};
  const float kR2[2] = {static_cast<float>(-152.94239396), static_cast<float>(59.28637943)};
  const float kG2[2] = {static_cast<float>(4.27729857), static_cast<float>(2.82956604)};
  const float kB2[2] = {static_cast<float>(-89.90310912), static_cast<float>(27.34824973)};
  x = clampf(x, 0.f, 1.f);
  const float v4[4] = {1.0f, x, x * x, x * x * x};
  const float v2[2] = {v4[2] * v4[2], v4[3] * v4[2]};
  // glm::dot: vec4 (x + y) + (z + w) of the products, vec2 x + y
  auto dot4 = [&](const float* k) { return (v4[0] * k[0] + v4[1] * k[1]) + (v4[2] * k[2] + v4[3] * k[3]); };
  auto dot2 = [&](const float* k) { return v2[0] * k[0] + v2[1] * k[1]; };
  return f3{dot4(kR4) + dot2(kR2), dot4(kG4) + dot2(kG2), dot4(kB4) + dot2(kB2)};
}

__global__ void __launch_bounds__(256)
heatmap_kernel(const DScene g, const RenderArgs A, float factor, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const Lds L = stage_lds(g, A, (VIMG_LDS unsigned char*)lds_raw);
  const uint32_t W = static_cast<uint32_t>(g.res_x), H = static_cast<uint32_t>(g.res_y);
  const uint32_t item = blockIdx.x * 256u + threadIdx.x;
  if (item >= A.num_local_tiles * 64u) return;
  // work items are tile-major: the reference's 8x8 tiles in its x-major order, shard r of n owns
  // tiles t with t % n == r (as render_kernel)
  const uint32_t tile = (item >> 6) * A.tile_world + A.tile_rank;
  const uint32_t within = item & 63u;
  const uint32_t tx = tile / A.tiles_y, ty = tile - tx * A.tiles_y;
  const uint32_t px = tx * 8 + (within & 7u), py = ty * 8 + (within >> 3);
  if (tx >= A.tiles_x || px >= W || py >= H) return;

  const uint64_t image_index = uint64_t(px) + uint64_t(H - 1 - py) * W;
  Rng rng{0};
  pcg_seed(rng, image_index);
  float pixel_hit_accumulator = 0.f;
  for (uint32_t smp = 0; smp < A.samples; ++smp) {
    const f2 off = random_x_y_r2(px + py + smp);
    const float rand2 = rand_float(rng);   // right-to-left argument evaluation
    const float rand1 = rand_float(rng);
    TravRay ray;
    generate_ray(g, static_cast<float>(px) + off.x, static_cast<float>(py) + off.y, rand1, rand2,
                 ray.o, ray.d);
    ray.min_t = 0.0001f;
    ray.max_t = VIMG_INF;
    Counters cnt{0, 0, 0, 0, 0, 0, 0, 0};
    HitRec rec;
    traverse<false>(g, L, ray, rec, cnt, true);
    // root 0.5 (counted before the root test's outcome is looked at) + 2 x 0.5 per visited
    // internal node + 1 per primitive test
    pixel_hit_accumulator += 0.5f + static_cast<float>(cnt.internal) + static_cast<float>(cnt.prim);
  }
  const float v = static_cast<float>(
      static_cast<uint32_t>(pixel_hit_accumulator / static_cast<float>(A.samples)));
  const f3 col = turbo_colormap(v / factor);
  const size_t o = (A.tile_world == 1) ? size_t(image_index) * 3 : size_t(item) * 3;
  out[o + 0] = col.x;
  out[o + 1] = col.y;
  out[o + 2] = col.z;
}

}  // namespace vimg
